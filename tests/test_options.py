"""pflare_amd.options: the reference's test command lines (tests/Makefile, quoted with their line numbers; $(KSP_RTOL) =
1e-10 in double precision, tests/Makefile:34) read as this package's generator / preconditioner / Krylov options.  CPU: the split; GPU: each
line run as the reference runs it - b = 0, x0 = 1 (tests/adv_diff_fd.c:179,325), converged within the line's own
-ksp_max_it (the reference's pass criterion: the driver fails unless KSPGetConvergedReason > 0) - and with the iteration
count of the C oracle on the same hierarchy."""
import numpy as np
import pytest

# (tests/Makefile lines, command line)
LINES = {
    "exact_2d_422": ("422-424", "./adv_diff_fd -u 0 -v 0 -alpha 1.0 -da_grid_x 8 -da_grid_y 8 -pc_type air "
                     "-pc_air_strong_threshold 0.0 -pc_air_a_drop 0.0 -pc_air_r_drop 0.0 -pc_air_inverse_type jacobi "
                     "-mg_coarse_ksp_type richardson -mg_coarse_ksp_max_it 10 -ksp_type richardson -ksp_norm_type unpreconditioned -ksp_max_it 1"),
    "upwind_3d_544": ("544-545", "./adv_diff_fd -dim 3 -da_grid_x 10 -da_grid_y 10 -da_grid_z 10 -pc_type air -ksp_pc_side right "
                      "-ksp_rtol 1e-10 -ksp_atol 1e-50 -pc_air_a_lump -ksp_max_it 4"),
    "newton_mf_16_549": ("549-550", "./adv_diff_fd -da_grid_x 5 -da_grid_y 5 -pc_type pflareinv -pc_pflareinv_type newton "
                         "-pc_pflareinv_matrix_free -pc_pflareinv_poly_order 16 -ksp_max_it 1"),
    "newton_mf_50_552": ("552-553", "./adv_diff_fd -da_grid_x 10 -da_grid_y 10 -pc_type pflareinv -pc_pflareinv_type newton "
                         "-pc_pflareinv_matrix_free -pc_pflareinv_poly_order 50 -ksp_max_it 1"),
    "auto_truncate_557": ("557-560", "./adv_diff_fd -da_grid_x 10 -da_grid_y 10 -ksp_type richardson -pc_type air "
                          "-pc_air_coarsest_inverse_type newton -pc_air_coarsest_poly_order 10 -pc_air_coarsest_matrix_free_polys "
                          "-pc_air_auto_truncate_start_level 1 -pc_air_auto_truncate_tol 1e-2 -ksp_max_it 3 "
                          "-pc_air_a_drop 1e-3 -pc_air_inverse_type power"),
    "fc_smoothing_300": ("300", "./adv_diff_fd -da_grid_x 8 -da_grid_y 8 -pc_type air -ksp_max_it 3 -pc_air_smooth_type fc"),
    "neumann_398": ("398", "./adv_diff_fd -u 0 -v 0 -alpha 1.0 -da_grid_x 8 -da_grid_y 8 -pc_type air -pc_air_inverse_type neumann -ksp_max_it 5 -pc_air_a_drop 1e-3"),
    "neumann_mf_400": ("400-401", "./adv_diff_fd -u 0 -v 0 -alpha 1.0 -da_grid_x 8 -da_grid_y 8 -pc_type air -pc_air_inverse_type neumann "
                       "-pc_air_matrix_free_polys -ksp_max_it 5 -pc_air_a_drop 1e-3"),
    "adv_1d_538": ("538-540", "./adv_1d -n 1000 -ksp_rtol 1e-10 -ksp_atol 1e-50 -ksp_pc_side right "
                   "-pc_air_coarsest_inverse_type newton -pc_air_coarsest_poly_order 10 -pc_air_coarsest_matrix_free_polys -ksp_max_it 2 "
                   "-pc_air_a_drop 1e-3 -pc_air_inverse_type power"),
    "improve_w_708": ("708-709", "./adv_diff_fd -da_grid_x 50 -da_grid_y 50 -pc_type air -pc_air_one_point_classical_prolong 0 "
                      "-pc_air_improve_w_its 1 -ksp_max_it 4	-pc_air_a_drop 1e-3 -pc_air_inverse_type power"),
    "newton_order_3_715": ("713-717, order 3", "./adv_diff_fd -da_grid_x 50 -da_grid_y 50 -pc_type air -pc_air_inverse_type newton -pc_air_poly_order 3 "
                           "-pc_air_inverse_sparsity_order 3 -ksp_norm_type unpreconditioned -ksp_max_it 6"),
    "lump_drop_100_1123": ("1121-1124, size 100", "./adv_diff_fd -da_grid_x 100 -da_grid_y 100 -pc_type air -ksp_pc_side right "
                           "-ksp_rtol 1e-10 -ksp_atol 1e-50 -pc_air_a_lump -pc_air_a_drop 1e-5 -pc_air_strong_threshold 0.99 -ksp_max_it 6"),
}


def test_the_split_of_a_line():
    from pflare_amd.options import parse_petsc_args
    o = parse_petsc_args(LINES["exact_2d_422"][1])
    assert o["pc_type"] == "air"
    assert o["gen"] == dict(u=0.0, v=0.0, alpha=1.0, M=8, N=8)
    assert o["pc"] == dict(strong_threshold=0.0, a_drop=0.0, r_drop=0.0, inverse_type="jacobi", mg_coarse_ksp_max_it=10)
    assert o["ksp"] == dict(ksp_type="richardson", norm_type="unpreconditioned", max_it=1)
    assert o["other"] == dict(mg_coarse_ksp_type="richardson")      # nothing is dropped silently
    o = parse_petsc_args(LINES["upwind_3d_544"][1])
    assert o["gen"] == dict(dim=3, M=10, N=10, P=10)
    assert o["pc"] == dict(a_lump=True)                             # a flag without a value
    assert o["ksp"] == dict(pc_side="right", rtol=1e-10, atol=1e-50, max_it=4)
    o = parse_petsc_args(LINES["newton_mf_16_549"][1])
    assert o["pc_type"] == "pflareinv" and o["pc"] == dict(type="newton", matrix_free=True, poly_order=16)
    o = parse_petsc_args(LINES["auto_truncate_557"][1])
    assert o["pc"] == dict(coarsest_inverse_type="newton", coarsest_poly_order=10, coarsest_matrix_free_polys=True,
                           auto_truncate_start_level=1, auto_truncate_tol=1e-2, a_drop=1e-3, inverse_type="power")
    # negative numbers are values, not options; a list of tokens is taken as it is
    o = parse_petsc_args(["-u", "-1.5", "-pc_air_a_lump", "false", "-ksp_gmres_restart", "7", "-log_view"])
    assert o["gen"] == dict(u=-1.5) and o["pc"] == dict(a_lump=False) and o["ksp"] == dict(restart=7) and o["other"] == dict(log_view=True)


def test_build_gives_the_objects_without_a_device():
    import pflare_amd as P
    from pflare_amd.options import build
    pc, ksp, gen, other = build(LINES["lump_drop_100_1123"][1])
    assert isinstance(pc, P.PCAIR) and isinstance(ksp, P.KSP) and gen == dict(M=100, N=100) and other == {}
    assert ksp.o.max_it == 6 and ksp.o.pc_side == 1 and ksp.o.rtol == 1e-10
    pc, _, _, _ = build(LINES["newton_mf_50_552"][1])
    assert isinstance(pc, P.PCPFLAREINV)
    o = __import__("pflare_amd.options", fromlist=["parse_petsc_args"]).parse_petsc_args(LINES["adv_1d_538"][1])
    assert o["program"] == "adv_1d" and o["pc_type"] == "air" and o["gen"] == dict(n=1000)      # tests/adv_1d.c:142 sets the type
    from pflare_amd.options import matrix
    assert matrix(LINES["adv_1d_538"][1]).shape == (1000, 1000) and matrix(LINES["fc_smoothing_300"][1]).shape == (64, 64)
    with pytest.raises(ValueError):
        build("./adv_diff_fd -pc_type hypre")
    with pytest.raises(P.AirhipError):           # an option this build does not have is refused by the library, not ignored
        build("./adv_diff_fd -pc_type air -pc_air_no_such_option 3")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LINES))
def test_reference_line_converges_as_the_reference_requires(gpu, name):
    import torch
    import pflare_amd as P
    from pflare_amd.options import build
    from oracle import c_oracle
    from pflare_amd.options import matrix
    pc, ksp, gen, other = build(LINES[name][1])
    A = matrix(LINES[name][1])
    n = A.shape[0]
    pc.setup(A)
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    res = ksp.solve(b, x, pc=pc, A=P.DeviceMat(A))
    assert res["reason"] > 0, (name, res)                      # converged within the line's -ksp_max_it
    if isinstance(pc, P.PCAIR):
        o = ksp.o
        ch = c_oracle.CHierarchy(pc)
        from pflare_amd.options import parse_petsc_args
        coarse_its = parse_petsc_args(LINES[name][1])["pc"].get("mg_coarse_ksp_max_it", 0)
        if coarse_its:
            ch.set_coarse_richardson(coarse_its)
        _, its, reason, rn, _ = ch.ksp(
            A, np.zeros(n), np.ones(n), {0: "richardson", 1: "gmres"}[o.type], {0: "left", 1: "right"}[o.pc_side],
            {0: "preconditioned", 1: "unpreconditioned"}[o.norm_type], rtol=o.rtol, atol=o.atol, divtol=o.dtol, max_it=o.max_it,
            restart=o.restart)
        assert (res["its"], res["reason"]) == (its, reason), (name, res, its, reason)
