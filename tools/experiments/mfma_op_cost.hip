// What does ONE instruction of a given kind cost a wave that keeps the matrix pipe busy?  (gfx950, v_mfma_f32_32x32x16_bf16)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/moc tools/experiments/mfma_op_cost.hip && /tmp/moc
//
// Loop body = 24 MFMAs on 4 independent accumulators, N instructions of the kind after every MFMA (independent chains), everything
// `asm volatile` (program order = listing).  One workgroup per CU, 1 or 2 waves per SIMD.  Printed per kind: wall time per iteration
// relative to the MFMA-only loop at N = 2 and N = 4 per MFMA, and the extra time per instruction in core cycles (at the clock the
// MFMA-only loop implies for 32 cycles per MFMA).  tools/experiments/mfma_valu_overlap.hip showed that up to 6 v_fmac_f32 per MFMA are
// free and that v_pk_fma_f32 is not: this is the table for the other instructions the split kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float sp2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

enum Kind { FMAC, PK_FMA, PK_ADD, PK_MUL, CVT_PK_BF16, LSHL_ADD_U64, MAD_U64_U32, CNDMASK, AND_B32, ADD_U32, ADD_F32_E64, EXP_F32, MOV_B32,
            ACC_WRITE, ACC_READ, READLANE, PERM_B32, MOV_DPP, DS_READ_B32, DS_READ2_B32, DS_READ_B128, DS_WRITE_B32, DS_WRITE_B128,
            GLOAD_DWORD, GLOAD_DWORDX4, FMA_F64, ADD3_U32, SUB_F32, FMAC_DEP, SPLIT_CHAIN, NKINDS };
static const char* kNames[] = {"v_fmac_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_lshl_add_u64", "v_mad_u64_u32",
                               "v_cndmask_b32", "v_and_b32", "v_add_u32", "v_add_f32_e64", "v_exp_f32", "v_mov_b32", "v_accvgpr_write_b32",
                               "v_accvgpr_read_b32", "v_readlane_b32", "v_perm_b32", "v_mov_b32_dpp", "ds_read_b32", "ds_read2_b32", "ds_read_b128",
                               "ds_write_b32", "ds_write_b128", "global_load_dword", "global_load_dwordx4", "v_fma_f64", "v_add3_u32", "v_sub_f32", "v_fmac_f32, ONE dependent chain", "cvt_pk/lshl/and/sub chain of the 3-way split (dependent)"};

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))

template <int KIND>
__device__ __forceinline__ void one(float& f, sp2& q, unsigned& u, unsigned long long& w, double& dd, u32x4& l4, u32x2& l2, unsigned laddr,
                                    const float* gp, float s0, float s1, sp2 p0, sp2 p1) {
  if (KIND == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f) : "v"(s0), "v"(s1));
  if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(p0), "v"(p1));
  if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(q) : "v"(p1));
  if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(q) : "v"(p0));
  if (KIND == CVT_PK_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(s0), "v"(s1));
  if (KIND == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w) : "v"(w));
  if (KIND == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(u), "v"(laddr) : "vcc");
  if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(laddr));
  if (KIND == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(laddr));
  if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(laddr));
  if (KIND == ADD_F32_E64) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(f) : "v"(s1));
  if (KIND == EXP_F32) asm volatile("v_exp_f32 %0, %1" : "=v"(f) : "v"(s1));
  if (KIND == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(laddr));
  if (KIND == ACC_WRITE) asm volatile("v_accvgpr_write_b32 a0, %0" : : "v"(laddr) : "a0");
  if (KIND == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(u) : : "a0");
  if (KIND == READLANE) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(laddr) : "s20");
  if (KIND == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(laddr), "v"(u));
  if (KIND == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u) : "v"(laddr));
  if (KIND == DS_READ_B32) asm volatile("ds_read_b32 %0, %1" : "=v"(u) : "v"(laddr));
  if (KIND == DS_READ2_B32) asm volatile("ds_read2_b32 %0, %1 offset1:81" : "=v"(l2) : "v"(laddr));
  if (KIND == DS_READ_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(l4) : "v"(laddr));
  if (KIND == DS_WRITE_B32) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(u));
  if (KIND == DS_WRITE_B128) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(l4));
  if (KIND == GLOAD_DWORD) asm volatile("global_load_dword %0, %1, %2" : "=v"(u) : "v"(laddr), "s"(gp));
  if (KIND == GLOAD_DWORDX4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(l4) : "v"(laddr), "s"(gp));
  if (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(dd) : "v"(w));
  if (KIND == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u) : "v"(laddr));
  if (KIND == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f) : "v"(s1));
  if (KIND == FMAC_DEP) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f) : "v"(s0), "v"(s1));
}

template <int KIND, int N>
__global__ __launch_bounds__(512) void cost_kernel(float* out, const float* gp, int iters) {
  __shared__ uint4 lds[4096];
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0, threadIdx.x));
  const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0, threadIdx.x, 0));
  float f0 = 0.f, f1 = 1.f, f2 = 2.f, f3 = 3.f;
  sp2 q0 = {0.f, 1.f}, q1 = {1.f, 1.f}, q2 = {2.f, 1.f}, q3 = {3.f, 1.f};
  unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
  unsigned long long w0 = threadIdx.x, w1 = 1, w2 = 2, w3 = 3;
  double d0 = 0., d1 = 1., d2 = 2., d3 = 3.;
  u32x4 l0 = {0, 0, 0, 0}, l1 = l0, l2_ = l0, l3 = l0;
  u32x2 h0 = {0, 0}, h1 = h0, h2 = h0, h3 = h0;
  const float s0 = 1.0001f, s1 = 1e-6f * threadIdx.x;
  const sp2 p0 = {1.0001f, 0.5f}, p1 = {1e-6f, 1e-7f};
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_uint4(i, 0, 0, 0);
  __syncthreads();
  const unsigned laddr = (unsigned)(threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 4096u;
#define OPS()                                                                      \
  if (KIND == SPLIT_CHAIN) {                                                           \
    if (N > 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(f0), "v"(f1));                 \
    if (N > 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u1) : "v"(u0));                               \
    if (N > 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f0) : "v"(u1));                                   \
    if (N > 3) asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %2, %2, %0" : "=&v"(u2), "+v"(u0), "+v"(f1)); \
  } else {                                                                             \
  if (N > 0) one<KIND>(f0, q0, u0, w0, d0, l0, h0, laddr, gp, s0, s1, p0, p1);        \
  if (N > 1) one<KIND>(KIND == FMAC_DEP ? f0 : f1, q1, u1, w1, d1, l1, h1, laddr, gp, s0, s1, p0, p1);        \
  if (N > 2) one<KIND>(KIND == FMAC_DEP ? f0 : f2, q2, u2, w2, d2, l2_, h2, laddr, gp, s0, s1, p0, p1);       \
  if (N > 3) one<KIND>(KIND == FMAC_DEP ? f0 : f3, q3, u3, w3, d3, l3, h3, laddr, gp, s0, s1, p0, p1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      MFMA(acc0); OPS() MFMA(acc1); OPS() MFMA(acc2); OPS() MFMA(acc3); OPS()
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  s += f0 + f1 + f2 + f3 + q0[0] + q1[1] + q2[0] + q3[1] + (float)(u0 + u1 + u2 + u3) + (float)(w0 + w1 + w2 + w3) + (float)(d0 + d1 + d2 + d3);
  s += (float)(l0[0] + l1[1] + l2_[0] + l3[1] + h0[0] + h1[1] + h2[0] + h3[1]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int N>
double run(int threads, float* out, const float* gp) {
  const int iters = 300, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  cost_kernel<KIND, N><<<blocks, threads>>>(out, gp, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  cost_kernel<KIND, N><<<blocks, threads>>>(out, gp, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms * 1e-3 / iters;
}

template <int KIND>
void row(float* out, const float* gp, const double base[2]) {
  for (int wi = 0; wi < 2; ++wi) {
    const int threads = 256 << wi;
    const double t2 = run<KIND, 2>(threads, out, gp), t4 = run<KIND, 4>(threads, out, gp);
    const double cyc = base[wi] / (24.0 * 32.0 * (wi + 1));  // seconds per core cycle implied by the MFMA-only loop
    // per-instruction cost: extra time over the MFMA-only loop / instructions issued by ONE wave (24 * N); for 2 waves per SIMD both waves' instructions share the SIMD
    printf("  %-22s %d wave(s)/SIMD:  N=2 %.3f x  (+%.1f cycles per instr)   N=4 %.3f x  (+%.1f cycles per instr)\n", kNames[KIND], wi + 1,
           t2 / base[wi], (t2 - base[wi]) / cyc / (48.0 * (wi + 1)), t4 / base[wi], (t4 - base[wi]) / cyc / (96.0 * (wi + 1)));
  }
}

template <int K>
void rows(float* out, const float* gp, const double base[2]) {
  row<K>(out, gp, base);
  if constexpr (K + 1 < NKINDS) rows<K + 1>(out, gp, base);
}

int main() {
  float *out, *gp;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&gp, 1 << 20);
  hipMemset(gp, 0, 1 << 20);
  double base[2];
  base[0] = run<FMAC, 0>(256, out, gp);
  base[1] = run<FMAC, 0>(512, out, gp);
  printf("MFMA-only loop (24 MFMAs): %.3f us with 1 wave per SIMD, %.3f us with 2  (768 / 1536 cycles: %.2f / %.2f GHz)\n", base[0] * 1e6, base[1] * 1e6,
         768.0 / base[0] * 1e-9, 1536.0 / base[1] * 1e-9);
  rows<0>(out, gp, base);
  return 0;
}
