#!/usr/bin/env python3
"""Reads the public enums of the reference's include/pflare.h (names and values only) into
tests/golden/reference_enums.json.  Run in the build container, where /root/reference exists:
    python tests/golden/make_reference_enums.py [/root/reference]
The fixture is data (enumerator names and their integer values), not source text."""
import json
import os
import re
import sys


def parse_enums(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        vals, nxt = {}, 0
        for item in [t.strip() for t in body.split(",") if t.strip()]:
            m = re.match(r"(\w+)\s*(?:=\s*(-?\d+))?$", item)
            if m.group(2) is not None:
                nxt = int(m.group(2))
            vals[m.group(1)] = nxt
            nxt += 1
        out[name] = vals
    return out


def parse_prototypes(path):
    """{function name: [parameter types]} of the PETSC_EXTERN PetscErrorCode prototypes"""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for name, params in re.findall(r"PETSC_EXTERN\s+PetscErrorCode\s+(\w+)\s*\(([^)]*)\)\s*;", txt):
        out[name] = [re.sub(r"\s+", " ", q.strip()).replace("const char *", "const char*") for q in params.split(",")]
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    enums = parse_enums(os.path.join(ref, "include", "pflare.h"))
    here = os.path.dirname(os.path.abspath(__file__))
    json.dump({"source": "include/pflare.h of the reference (enumerator names and values)", "enums": enums},
              open(os.path.join(here, "reference_enums.json"), "w"), indent=1, sort_keys=True)
    print({k: len(v) for k, v in enums.items()})
    protos = parse_prototypes(os.path.join(ref, "include", "pflare.h"))
    json.dump({"source": "include/pflare.h of the reference (names and parameter types of its PETSC_EXTERN PetscErrorCode prototypes)",
               "prototypes": protos}, open(os.path.join(here, "reference_prototypes.json"), "w"), indent=0, sort_keys=True)
    print(len(protos), "prototypes")
