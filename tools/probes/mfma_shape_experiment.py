"""Timing-only experiment for VERDICT r3 item 3(i): does v_mfma_f32_32x32x16_bf16 (half the operand-register reads per flop of the
16x16x32 the selection kernel uses) make the ping-pong main loop faster under the board's power cap?

Builds THREE variant libraries from patched COPIES of csrc/sweep_gemm_bf16.hip (tools/probes/out/, never the product source):
  noepi_16   the ping-pong kernel with its epilogue compiled out (quick test + protocol): the main loop alone
  noepi_32   the same, every phase's 16 x (16x16x32) products replaced by 8 x (32x32x16) on 16-register accumulators —
             the same fragment reads, LDS-DMA requests, barriers and matrix-pipe cycles; operands are whatever the fragment
             registers hold (results are garbage: TIMING ONLY)
  full       unpatched (reference point)
and prints how to run them: VELESDB_HIP_LIB=<lib> python tools/bf16_probe.py --rows 4000000."""
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(R, "tools", "probes", "out")
src = open(os.path.join(R, "velesdb_amd", "csrc", "sweep_gemm_bf16.hip")).read()
a = src.index("template <int METRIC, bool I8 = false>\n__global__ __launch_bounds__(512, 2) void sweep_topk_gemm_bf16_pp")
head, pp = src[:a], src[a:]

# (P1) no epilogue inside the ping-pong kernel
pp_noepi = re.sub(r'#define VDB_G16_ACC_F\(V\).*?#undef VDB_G16_ACC_F\n', '''    const bool last = !more; (void)last;
    if (wr == 0) pp_barrier();
''', pp, count=1, flags=re.S)
assert pp_noepi != pp

# (P2) 32x32x16 products on 16-register accumulators
pp32 = pp_noepi.replace("  f32x4 acc[8][4];\n  f32x4 bv[4][2], a0v[4][2], a1v[4][2];", "  f32x16 acc16[8];\n  f32x4 bv[4][2], a0v[4][2], a1v[4][2];")
assert pp32 != pp_noepi
m = re.search(r'#define VDB_PP_MFMA\(AV, RF0, T0, FIRST\) do \{.*?\} while \(0\)\n', pp32, flags=re.S)
new_mfma = '''#define VDB_PP_MFMA(AV, RF0, T0, FIRST) do { \\
_Pragma("unroll") \\
    for (int s_ = 0; s_ < 4; s_++) \\
_Pragma("unroll") \\
      for (int h_ = 0; h_ < 2; h_++) { \\
        f32x16& c_ = acc16[((RF0) / 4) * 4 + ((T0) / 2) * 2 + h_]; \\
        if ((FIRST) && s_ == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(c_) : "v"(AV[2 * h_ + (s_ & 1)][s_ >> 1]), "v"(bv[(T0) + (s_ & 1)][s_ >> 1])); \\
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c_) : "v"(AV[2 * h_ + (s_ & 1)][s_ >> 1]), "v"(bv[(T0) + (s_ & 1)][s_ >> 1])); \\
      } \\
  } while (0)
'''
pp32 = pp32[:m.start()] + new_mfma + pp32[m.end():]
head32 = head.replace("typedef float f32x4 __attribute__((ext_vector_type(4)));", "typedef float f32x4 __attribute__((ext_vector_type(4)));\ntypedef float f32x16 __attribute__((ext_vector_type(16)));")

FL = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -w".split()
objs = [os.path.join(R, "velesdb_amd", "lib", "obj", f) for f in os.listdir(os.path.join(R, "velesdb_amd", "lib", "obj")) if f.endswith(".o") and f != "sweep_gemm_bf16.o"]
for name, text in (("noepi_16", head + pp_noepi), ("noepi_32", head32 + pp32)):
    d = os.path.join(O, name)
    os.makedirs(d, exist_ok=True)
    # the copy includes the product's headers / .inc files from csrc/
    cp = os.path.join(R, "velesdb_amd", "csrc", f"_exp_{name}.hip")
    open(cp, "w").write(text)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FL, "-c", cp, "-o", os.path.join(d, "g16.o")])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(O, f"libvelesdb_hip_{name}.so"), *objs, os.path.join(d, "g16.o")])
    finally:
        os.remove(cp)
    print("built", name)
