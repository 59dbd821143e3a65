/* noreplace_probe.c - does mmap(MAP_FIXED_NOREPLACE) refuse a range that is (partly) mapped on this kernel?  (diagnostic) */
#define _GNU_SOURCE
#include <stdio.h>
#include <string.h>
#include <errno.h>
#include <sys/mman.h>
#include <sys/utsname.h>
#ifndef MAP_FIXED_NOREPLACE
#define MAP_FIXED_NOREPLACE 0x100000
#endif
int main(void) {
  struct utsname u; uname(&u); printf("kernel %s %s\n", u.sysname, u.release);
  const size_t big = (size_t)1 << 30, small = (size_t)64 << 20;
  char* hole = mmap(NULL, big, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  munmap(hole, big);                                             /* a free range we know the address of */
  char* h = mmap(hole + small, small, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
  printf("array inside the free range: asked %p got %p\n", (void*)(hole + small), (void*)h);
  memset(h, 7, small);
  errno = 0;
  char* t = mmap(hole, big, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE, -1, 0);
  printf("cover over the whole range: got %p (asked %p), errno %d (%s)\n", (void*)t, (void*)hole, errno, strerror(errno));
  if (t == (char*)hole) printf("=> MAP_FIXED_NOREPLACE REPLACED a live mapping on this kernel\n");
  else printf("=> refused (or placed elsewhere): the live mapping is intact, first byte %d\n", h[0]);
  return 0;
}
