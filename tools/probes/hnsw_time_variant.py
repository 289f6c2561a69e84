#!/usr/bin/env python3
"""Builds tools/probes/out/libvelesdb_hip_hnswprobe.so: hnsw_kernels.hip with wall-clock stamps around the three legs of a layer-0
expansion (pop + neighbour ids | barrier + rows + distances | admission); query 0 of every launch prints the averages."""
import os, re, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = open(R + '/velesdb_amd/csrc/hnsw_kernels.hip').read()
def rep(old, new):
    global t
    assert old in t, old
    t = t.replace(old, new, 1)
rep('    uint32_t n_dist = 0, n_expand = 0, logn = 0, overflow = 0, m_prev = 0, rr_m = 0;',
    '    uint32_t n_dist = 0, n_expand = 0, logn = 0, overflow = 0, m_prev = 0, rr_m = 0;\n    unsigned long long tA = 0, tB = 0, tC = 0, tC1 = 0, tC2 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0; uint32_t nadm = 0, nbatch = 0;')
rep('            const uint32_t idx = list.first_unexpanded(lane);', '            c0 = wall_clock64();\n            const uint32_t idx = list.first_unexpanded(lane);')
rep('                uint32_t nc = rfl(ncv);\n                nc = min(nc, lim);', '                uint32_t nc = rfl(ncv);\n                nc = min(nc, lim);\n                { volatile uint32_t sink = rfl(nb0); (void)sink; }\n                c1 = wall_clock64(); tA += c1 - c0;')
rep('          } else if (phase == P_Z_ADMIT) {', '          } else if (phase == P_Z_ADMIT) {\n            c2 = wall_clock64(); tB += c2 - c1;')
rep('              if (LAT && a.lat_spec) mask &= spec_mask;\n', '              if (LAT && a.lat_spec) mask &= spec_mask;\n              nadm += (uint32_t)__popcll(mask);\n              c3 = wall_clock64(); tC1 += c3 - c2;\n')
rep('                mask = 0ull;\n', '                { mask = 0ull; nbatch++; }\n              tC2 += wall_clock64() - c3;\n')
i = t.index('            phase = P_Z_POP;\n          } else if (phase == P_FINISH) {')
t = t[:i] + '            tC += wall_clock64() - c2;\n' + t[i:]
j = [m.start() for m in re.finditer(re.escape('atomicAdd(&a.stats[1], (unsigned long long)n_expand);'), t)][1]
k = t.index('\n', j)
t = t[:k] + '\n          if (qi == 0) printf("probe: expansions %u  pop+ids %.2f us  barrier+rows+dist %.2f us  admit %.2f us (prologue %.2f, batch %.2f) (per expansion), total %.1f us; admitted %u, batch chunks %u\\n", n_expand, tA * 0.01 / n_expand, tB * 0.01 / n_expand, tC * 0.01 / n_expand, tC1 * 0.01 / n_expand, tC2 * 0.01 / n_expand, (tA + tB + tC) * 0.01, nadm, nbatch);' + t[k:]
O = R + '/tools/probes/out'
os.makedirs(O, exist_ok=True)
open(O + '/hnsw_kernels_probe.hip', 'w').write(t)
FL = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w", "-I" + R + "/velesdb_amd/csrc", "-I" + R + "/include"]
subprocess.check_call(["/opt/rocm/bin/hipcc", *FL, "-c", O + '/hnsw_kernels_probe.hip', "-o", O + '/hnsw_kernels_probe.o'])
objs = [R + '/velesdb_amd/lib/obj/' + f for f in sorted(os.listdir(R + '/velesdb_amd/lib/obj')) if f != 'hnsw_kernels.o']
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", O + '/libvelesdb_hip_hnswprobe.so', *objs, O + '/hnsw_kernels_probe.o'])
os.remove(O + '/hnsw_kernels_probe.o')
print('built')
