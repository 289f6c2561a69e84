#!/usr/bin/env python3
"""Where do seed_scores_bf16's 40 us go?  Builds tools/probes/out/libvelesdb_hip_seed_{noloop,noepi,nostore}.so from patched copies of
sweep_split.hip: the k-loop compiled out / the epilogue reduced to one store per lane / the key stores dropped (timing only: results
are garbage).  Run each under tools/probes/r03_h.sh (VELESDB_HIP_LIB) and read the kernel's duration off the timeline."""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(R + '/velesdb_amd/csrc/sweep_split.hip').read()
O = R + '/tools/probes/out'
os.makedirs(O, exist_ok=True)
FL = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w", "-I" + R + "/velesdb_amd/csrc", "-I" + R + "/include"]
loop = '  for (uint32_t k0 = 0; k0 < dim; k0 += 64) {  // dim % 32 == 0: steps past dim are skipped'
store = '    keys[(size_t)q * ngrp + (row0 / 64u) * 4u + kk] = best;'
epi_a = '''        const float sc = finish_score<METRIC>(acc[rb][t][r], qn, METRIC == kCosine ? norms[row] : 1.0f);'''
assert loop in src and store in src and epi_a in src
variants = {
    'noloop': src.replace(loop, '  for (uint32_t k0 = 0; k0 < (dim == 0xFFFFFFFFu ? dim : 0u); k0 += 64) {'),
    'nostore': src.replace(store, '    if (best == 12345ull) keys[(size_t)q * ngrp + (row0 / 64u) * 4u + kk] = best;'),
    'nonorms': src.replace(epi_a, '        const float sc = finish_score<METRIC>(acc[rb][t][r], qn, 1.0f);'),
}
for name, t in variants.items():
    p = O + '/sweep_split_%s.hip' % name
    open(p, 'w').write(t)
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FL, "-c", p, "-o", O + '/ss_%s.o' % name])
    objs = [R + '/velesdb_amd/lib/obj/' + f for f in sorted(os.listdir(R + '/velesdb_amd/lib/obj')) if f != 'sweep_split.o']
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", O + '/libvelesdb_hip_seed_%s.so' % name, *objs, O + '/ss_%s.o' % name])
    os.remove(O + '/ss_%s.o' % name); os.remove(p)
    print('built', name)
