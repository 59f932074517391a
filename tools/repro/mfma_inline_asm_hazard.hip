// Reproducer for the "stale value in lanes 48-63 of one accumulator register" nondeterminism of round 1.
//
// Claim: on gfx950 (hipcc / ROCm 7.2) a vector instruction written as INLINE ASM that reads a VGPR an in-flight MFMA is still
// writing gets no hazard padding -- LLVM's GCNHazardRecognizer pads MFMA-result reads for instructions it models, not for the
// operands of an inline-asm blob -- and reads the previous contents in the lanes of the passes that have not retired yet (the
// last pass of a 32x32 MFMA writes lanes 48-63).  The same read through a compiler-visible instruction is padded and exact.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hazard tools/repro/mfma_inline_asm_hazard.hip && /tmp/hazard
// prints, per variant, how many of the 64 lanes x 16 registers differ from the reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ void k(const _Float16* a, const _Float16* b, float* out) {
  const int lane = threadIdx.x;
  h8 av, bv;
  for (int j = 0; j < 8; ++j) {
    av[j] = a[lane * 8 + j];
    bv[j] = b[lane * 8 + j];
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = -1.0f;  // what a stale read would return
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  float res[16];
  if (VARIANT == 0) {  // compiler-visible consumer: padded by the hazard recognizer
    for (int r = 0; r < 16; ++r) res[r] = acc[r] * 1.0f + 0.0f;
  } else if (VARIANT == 1) {  // inline-asm consumer straight after the MFMA
    for (int r = 0; r < 16; ++r) asm volatile("v_mov_b32 %0, %1" : "=&v"(res[r]) : "v"(acc[r]));
  } else {  // inline asm behind an explicit wait: 16 passes of the 32x32x16 MFMA + margin
    // the wait names an MFMA result register as an operand: a register-only MFMA is not ordered by a "memory" clobber and
    // hipcc would otherwise sink it below the nops
    asm volatile("s_nop 15\n s_nop 7" : "+v"(acc[15]));
    for (int r = 0; r < 16; ++r) asm volatile("v_mov_b32 %0, %1" : "=&v"(res[r]) : "v"(acc[r]));
  }
  for (int r = 0; r < 16; ++r) out[r * 64 + lane] = res[r];
}

int main() {
  std::vector<_Float16> a(512), b(512);
  for (int i = 0; i < 512; ++i) {
    a[i] = (_Float16)(0.25f * ((i * 7) % 11 - 5));
    b[i] = (_Float16)(0.5f * ((i * 5) % 13 - 6));
  }
  _Float16 *da, *db;
  float* dout;
  hipMalloc(&da, 1024);
  hipMalloc(&db, 1024);
  hipMalloc(&dout, 3 * 4096);
  hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
  std::vector<float> ref(1024), got(1024);
  int rc = 0;
  for (int v = 0; v < 3; ++v) {
    int worst = 0;
    for (int rep = 0; rep < 200; ++rep) {
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, da, db, dout);
      if (v == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, da, db, dout + 1024);
      if (v == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, da, db, dout + 2048);
      hipMemcpy(got.data(), dout + 1024 * v, 4096, hipMemcpyDeviceToHost);
      if (v == 0 && rep == 0) ref = got;
      int bad = 0, bad_hi = 0;
      for (int i = 0; i < 1024; ++i)
        if (got[i] != ref[i]) {
          ++bad;
          if ((i & 63) >= 48) ++bad_hi;
        }
      if (bad > worst) worst = bad;
      if (rep == 0)
        printf("variant %d (%s): %d of 1024 values differ from the padded read, %d of them in lanes 48-63\n", v,
               v == 0 ? "compiler-visible read" : (v == 1 ? "inline asm right after the MFMA" : "inline asm after s_nop"), bad, bad_hi);
    }
    printf("variant %d: worst over 200 launches: %d differing values\n", v, worst);
    if (v != 1 && worst) rc = 1;
  }
  return rc;
}
