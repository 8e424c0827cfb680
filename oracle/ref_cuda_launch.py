#!/usr/bin/env python3
"""Rewrite CUDA kernel-launch expressions `kernel<<<grid, block, ...>>>(args)` into `ref_no_cuda_launch("kernel", args)`
so that a reference .cu file whose CPU functions oracle/_ref needs can be compiled by g++ (see Makefile.ref).  The output
is a build product under oracle/_ref/gen/ (git-ignored); nothing but the launch expressions changes."""
import re
import sys

src = open(sys.argv[1]).read()
pat = re.compile(r"([A-Za-z_][A-Za-z_0-9]*(?:\s*<[^<>;(){}]*>)?)\s*<<<(.*?)>>>\s*\(", re.S)
out, n = pat.subn(lambda m: 'ref_no_cuda_launch("%s", ' % m.group(1).split("<")[0].strip(), src)
head = ('// GENERATED from %s by oracle/ref_cuda_launch.py: %d kernel launches stubbed; do not commit.\n'
        '#include <cstdio>\n#include <cstdlib>\n'
        'template <class... A> static void ref_no_cuda_launch(const char *k, A &&...) {\n'
        '  std::fprintf(stderr, "oracle/_ref: CUDA kernel %%s is not available on the CPU\\n", k);\n  std::abort();\n}\n'
        '#line 1 "%s"\n' % (sys.argv[1], n, sys.argv[1]))
open(sys.argv[2], "w").write(head + out)
print("%s: %d launches stubbed" % (sys.argv[1], n))
