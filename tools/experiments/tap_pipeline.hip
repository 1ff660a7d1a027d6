// The tap loop of sphere_fwd_split_kernel reduced to its LDS / vector / matrix skeleton (no global memory), to find out what a tap's
// ~2 700 cycles are made of and what a deeper software pipeline would buy before building it into the kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tap tools/experiments/tap_pipeline.hip && /tmp/tap
//
// 8 waves, one workgroup per CU.  Per tap a wave (a) samples 8 channels of its 32 pixels from a window in LDS (4 words + 4 FMAs each),
// splits them into 3 bf16 pieces and writes 3 x 1 KB to an operand buffer, (b) reads the 4 x 3 operand fragments of its row block
// (12 KB) and (c) issues 24 MFMAs.  MODE 0 = the shipped order: fragments of tap k are read at the top of tap k (two operand buffers);
// MODE 1 = three operand buffers: tap k multiplies fragments that were read during tap k - 1, reads those of tap k + 1 and samples tap
// k + 2 (needs 48 more registers);  MODE 2 = MODE 0 without the sampling (operand buffers constant);  MODE 3 = MODE 0 without the
// fragment reads (MFMAs on constant registers); MODE 4 = MFMAs + barrier only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WRP = 81, CP = 8 * WRP + 1, WIN = 16 * CP, OP = 8 * 3 * 64;

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = pack2(a, b);
  float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
  asm("" : "+v"(ra), "+v"(rb));
  p2 = pack2(ra, rb);
  float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
  asm("" : "+v"(sa), "+v"(sb));
  p3 = pack2(sa, sb);
}
__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE, int VPM = 4>
__global__ __launch_bounds__(512) void tap_kernel(float* out, int ntaps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* opbuf = reinterpret_cast<uint4*>(smem + ((WIN + WRP + 8 + 3) / 4) * 4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  for (int i = tid; i < WIN + WRP + 8; i += 512) smem[i] = 1.0f + 1e-3f * (i % 97);
  for (int i = tid; i < 3 * OP; i += 512) opbuf[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  __syncthreads();
  int roff[9];
  float4 rw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    roff[k] = (k % 3) * WRP + (k / 3) + (lane & 31);
    rw[k] = make_float4(0.25f + 0.01f * k, 0.25f, 0.25f - 0.01f * k, 0.25f);
  }
  uint4 a[3];
  for (int p = 0; p < 3; ++p) a[p] = make_uint4(0x3f803f80u + p, 0x3c003c00u, lane, 0x3f803f80u);
  f32x16 acc[4];
  for (int g = 0; g < 4; ++g)
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  const int gset = (wave / 4) * 4;

  auto sample = [&](int k, uint4* op) {
    const float* p = smem + half * 8 * CP + roff[k];
    const float4 tw = rw[k];
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = p + c * CP;
      v[c] = __builtin_fmaf(tw.w, q[WRP + 1], __builtin_fmaf(tw.z, q[1], __builtin_fmaf(tw.y, q[WRP], tw.x * q[0])));
      asm("" : "+v"(v[c]));
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
    uint4* dst = op + (wave * 3) * 64 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  };
  auto sample_read = [&](int k, float (&raw)[8][4]) {
    const float* p = smem + half * 8 * CP + roff[k];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = p + c * CP;
      raw[c][0] = q[0];
      raw[c][1] = q[WRP];
      raw[c][2] = q[1];
      raw[c][3] = q[WRP + 1];
    }
  };
  auto sample_finish = [&](int k, float (&raw)[8][4], uint4* op) {
    const float4 tw = rw[k];
    float v[8];
    uint32_t q1[4], q2[4], q3[4];
    if (MODE == 13) {  // no arithmetic at all: the words go out as they came
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q1[j] = __builtin_bit_cast(uint32_t, raw[2 * j][0]) ^ __builtin_bit_cast(uint32_t, raw[2 * j][1]);
        q2[j] = __builtin_bit_cast(uint32_t, raw[2 * j][2]) ^ __builtin_bit_cast(uint32_t, raw[2 * j][3]);
        q3[j] = __builtin_bit_cast(uint32_t, raw[2 * j + 1][0]) ^ __builtin_bit_cast(uint32_t, raw[2 * j + 1][3]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (MODE == 15) v[c] = raw[c][0] + raw[c][1] * raw[c][2] + raw[c][3];
        else v[c] = __builtin_fmaf(tw.w, raw[c][3], __builtin_fmaf(tw.z, raw[c][2], __builtin_fmaf(tw.y, raw[c][1], tw.x * raw[c][0])));
        asm("" : "+v"(v[c]));
      }
      if (MODE == 14) {  // bilinear combine only, no split
#pragma unroll
        for (int j = 0; j < 4; ++j) q1[j] = pack2(v[2 * j], v[2 * j + 1]), q2[j] = q1[j] + 1, q3[j] = q1[j] + 2;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      }
    }
    uint4* dst = op + (wave * 3) * 64 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  };
  auto frags = [&](const uint4* opr, uint4 (&bq)[4][3]) {
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[gi][p] = opr[((gset + gi) * 3 + p) * 64 + lane];
  };
#define TERM(BQ, PA, PB) _Pragma("unroll") for (int gi = 0; gi < 4; ++gi) acc[gi] = mfma(a[PA], BQ[gi][PB], acc[gi]);
#define TERMS(BQ) TERM(BQ, 2, 0) TERM(BQ, 0, 2) TERM(BQ, 1, 1) TERM(BQ, 1, 0) TERM(BQ, 0, 1) TERM(BQ, 0, 0)
#define SPREAD()                                            \
  _Pragma("unroll") for (int i = 0; i < 24; ++i) {          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      \
  }                                                         \
  __builtin_amdgcn_sched_barrier(0);

  uint4 bqc[4][3];
  for (int gi = 0; gi < 4; ++gi)
    for (int p = 0; p < 3; ++p) bqc[gi][p] = make_uint4(0x3f803f80u, lane, 0x3c003c00u, gi + p);

#define SPREAD2(NREADS)                                     \
  __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);   \
  _Pragma("unroll") for (int i = 0; i < 24; ++i) {          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      \
  }                                                         \
  __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);        \
  __builtin_amdgcn_sched_barrier(0);
  if (MODE == 5) {
    sample(0, opbuf);
    lds_barrier();
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        uint4 bq[4][3];
        float raw[8][4];
        frags(opbuf + ((t0 + i) & 1) * OP, bq);
        sample_read((i + 1) % 9, raw);
        TERMS(bq)
        sample_finish((i + 1) % 9, raw, opbuf + ((t0 + i + 1) & 1) * OP);
        SPREAD2(28)
        lds_barrier();
      }
    }
  } else if (MODE == 20 || MODE == 21 || MODE == 22) {
    // shipped two-buffer order with the fragments read piece by piece in the order the terms use them (0, 2, 1)
    sample(0, opbuf);
    lds_barrier();
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        uint4 bq[4][3];
        float raw[8][4];
        const uint4* opr = opbuf + ((t0 + i) & 1) * OP;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) bq[gi][0] = opr[((gset + gi) * 3 + 0) * 64 + lane];
        sample_read((i + 1) % 9, raw);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) bq[gi][2] = opr[((gset + gi) * 3 + 2) * 64 + lane];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) bq[gi][1] = opr[((gset + gi) * 3 + 1) * 64 + lane];
        TERMS(bq)
        sample_finish((i + 1) % 9, raw, opbuf + ((t0 + i + 1) & 1) * OP);
        if (MODE == 20) {
          __builtin_amdgcn_sched_group_barrier(0x100, 28, 0);
          _Pragma("unroll") for (int j = 0; j < 24; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        } else if (MODE == 21) {
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          }
          _Pragma("unroll") for (int j = 0; j < 16; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
          }
          _Pragma("unroll") for (int j = 0; j < 20; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
      }
    }
  } else if (MODE == 6) {
    uint4 bqA[4][3], bqB[4][3];
    sample(0, opbuf);
    sample(1, opbuf + OP);
    lds_barrier();
    frags(opbuf, bqA);
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; i += 2) {
        {
          float raw[8][4];
          sample_read((i + 2) % 9, raw);
          frags(opbuf + ((i + 1) % 3) * OP, bqB);
          TERMS(bqA)
          sample_finish((i + 2) % 9, raw, opbuf + ((i + 2) % 3) * OP);
          SPREAD2(28)
          lds_barrier();
        }
        {
          float raw[8][4];
          sample_read((i + 3) % 9, raw);
          frags(opbuf + ((i + 2) % 3) * OP, bqA);
          TERMS(bqB)
          sample_finish((i + 3) % 9, raw, opbuf + ((i + 3) % 3) * OP);
          SPREAD2(28)
          lds_barrier();
        }
      }
    }
  } else if (MODE == 7) {
    // one fragment set, refilled in place: piece 2 of the next tap after this tap's (0,2) term, piece 1 after the (1,1), (0,1) terms,
    // piece 0 at the top of the tap that uses it (its three terms come last)
    uint4 bq[4][3];
    sample(0, opbuf);
    sample(1, opbuf + OP);
    lds_barrier();
    frags(opbuf, bq);
    auto refill = [&](const uint4* opr, int p) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) bq[gi][p] = opr[((gset + gi) * 3 + p) * 64 + lane];
    };
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        float raw[8][4];
        const uint4* cur = opbuf + (i % 3) * OP;
        const uint4* nxt = opbuf + ((i + 1) % 3) * OP;
        if (i > 0 || t0 > 0) refill(cur, 0);
        sample_read((i + 2) % 9, raw);
        TERM(bq, 0, 2)
        refill(nxt, 2);
        TERM(bq, 1, 1) TERM(bq, 0, 1)
        refill(nxt, 1);
        TERM(bq, 2, 0) TERM(bq, 1, 0) TERM(bq, 0, 0)
        sample_finish((i + 2) % 9, raw, opbuf + ((i + 2) % 3) * OP);
        __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        _Pragma("unroll") for (int j = 0; j < 12; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
      }
    }
  } else if (MODE >= 8 && MODE <= 19) {
    // MODE 7 with the LDS reads SPREAD over the MFMA slots instead of batched at the top: slots 0..3 issue piece 0 of this tap (one
    // fragment each) and the window words of channels 0..3, slots 4..7 piece 2 of the next tap and channels 4..7, slots 12..15 piece 1
    // of the next tap; the arithmetic follows its words by two slots
    uint4 bq[4][3];
    sample(0, opbuf);
    sample(1, opbuf + OP);
    lds_barrier();
    frags(opbuf, bq);
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        float raw[8][4];
        const uint4* cur = opbuf + (i % 3) * OP;
        const uint4* nxt = opbuf + ((i + 1) % 3) * OP;
        const int k = (i + 2) % 9;
        const float* p = smem + half * 8 * CP + roff[k];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          if (MODE != 11 && MODE < 16) bq[gi][0] = cur[((gset + gi) * 3 + 0) * 64 + lane];
          const float* q = p + gi * CP;
          if (MODE != 12 && MODE < 16) raw[gi][0] = q[0], raw[gi][1] = q[WRP], raw[gi][2] = q[1], raw[gi][3] = q[WRP + 1];
          else raw[gi][0] = raw[gi][1] = raw[gi][2] = raw[gi][3] = acc[0][gi];
        }
        TERM(bq, 0, 2)
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          if (MODE != 11 && MODE < 16) bq[gi][2] = nxt[((gset + gi) * 3 + 2) * 64 + lane];
          const float* q = p + (4 + gi) * CP;
          if (MODE != 12 && MODE < 16) raw[4 + gi][0] = q[0], raw[4 + gi][1] = q[WRP], raw[4 + gi][2] = q[1], raw[4 + gi][3] = q[WRP + 1];
          else raw[4 + gi][0] = raw[4 + gi][1] = raw[4 + gi][2] = raw[4 + gi][3] = acc[1][gi];
        }
        TERM(bq, 1, 1) TERM(bq, 0, 1)
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) if (MODE != 11 && MODE < 16) bq[gi][1] = nxt[((gset + gi) * 3 + 1) * 64 + lane];
        TERM(bq, 2, 0) TERM(bq, 1, 0) TERM(bq, 0, 0)
        if (MODE == 17) { asm volatile("" :: "v"(raw[0][0]), "v"(raw[7][3])); } else sample_finish(k, raw, MODE == 10 || MODE == 16 ? reinterpret_cast<uint4*>(out) + (size_t)blockIdx.x * 3 * OP + ((i + 2) % 3) * OP : opbuf + ((i + 2) % 3) * OP);
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else lds_barrier();
      }
    }
  } else if (MODE == 1) {
    uint4 bqA[4][3], bqB[4][3];
    sample(0, opbuf);
    sample(1, opbuf + OP);
    lds_barrier();
    frags(opbuf, bqA);
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; i += 2) {
        // tap i: multiply A (read during tap i - 1), read B = fragments of i + 1, sample i + 2
        frags(opbuf + ((i + 1) % 3) * OP, bqB);
        sample((i + 2) % 9, opbuf + ((i + 2) % 3) * OP);
        TERMS(bqA)
        SPREAD()
        lds_barrier();
        frags(opbuf + ((i + 2) % 3) * OP, bqA);
        sample((i + 3) % 9, opbuf + ((i + 3) % 3) * OP);
        TERMS(bqB)
        SPREAD()
        lds_barrier();
      }
    }
  } else {
    if (MODE == 0 || MODE == 3) sample(0, opbuf);
    lds_barrier();
    for (int t0 = 0; t0 < ntaps; t0 += 18) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        uint4 bq[4][3];
        const uint4* opr = opbuf + ((t0 + i) & 1) * OP;
        uint4* opw = opbuf + ((t0 + i + 1) & 1) * OP;
        if (MODE == 0 || MODE == 2) frags(opr, bq);
        if (MODE == 0 || MODE == 3) sample((i + 1) % 9, opw);
        if (MODE == 0 || MODE == 2) { TERMS(bq) } else { TERMS(bqc) }
        SPREAD()
        lds_barrier();
      }
    }
  }
  float s = 0.f;
  for (int g = 0; g < 4; ++g)
    for (int r = 0; r < 16; ++r) s += acc[g][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(float* out, const char* what) {
  const size_t lds = (size_t)(((WIN + WRP + 8 + 3) / 4) * 4) * 4 + 3 * (size_t)OP * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(tap_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntaps = 720;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  tap_kernel<MODE><<<256, 512, lds>>>(out, 18);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  tap_kernel<MODE><<<256, 512, lds>>>(out, ntaps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_tap = ms * 1e3 / ntaps;
  printf("MODE %d  %-90s %.3f us per tap = %.0f cycles at 2.0 GHz (matrix pipe alone: 1536)  err %s\n", MODE, what, us_tap, us_tap * 2000.0,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 3 * OP * 16 + 256 * 512 * sizeof(float));
  run<4>(out, "24 MFMAs on constant registers + barrier");
  run<3>(out, "sampling + 24 MFMAs on constant registers (no fragment reads)");
  run<2>(out, "fragment reads at the top of the tap + 24 MFMAs (no sampling)");
  run<0>(out, "shipped order: fragment reads, sampling of the next tap, 24 MFMAs");
  run<1>(out, "three deep: MFMAs on fragments read one tap earlier, reads for the next tap, sampling two taps ahead");
  run<5>(out, "shipped order, all 28 LDS reads of the tap issued first, arithmetic 4 per MFMA, writes last");
  run<20>(out, "shipped order, fragments piece-major (0, 2, 1) around the window reads, all reads first");
  run<21>(out, "shipped order, 4 reads, then 8 x (MFMA + 3 reads), then 16 x (MFMA + 6 arithmetic)");
  run<22>(out, "shipped order, 4 reads, then 4 x (MFMA + 6 reads), then 20 x (MFMA + 5 arithmetic)");
  run<7>(out, "three deep with ONE fragment set refilled in place, reads batched, arithmetic 4 per MFMA");
  run<8>(out, "three deep, ONE fragment set, LDS reads spread over the MFMA slots, arithmetic 4 per MFMA");
  run<9>(out, "  the same without the per-tap barrier (results meaningless)");
  run<10>(out, "  the same with the operand stores going to global memory instead of LDS");
  run<11>(out, "  the same without the fragment reads");
  run<12>(out, "  the same without the window reads (arithmetic and stores kept)");
  run<13>(out, "  the same without any sampling arithmetic (window words stored as they are)");
  run<14>(out, "  the same with the bilinear combine but without the 3-way split");
  run<16>(out, "  MFMAs + sampling arithmetic + barrier only (no LDS instruction in the loop; stores to global memory)");
  run<17>(out, "  MFMAs + barrier only, in MODE 8's instruction order");
  run<6>(out, "three deep, all 28 LDS reads of the tap issued first (sampling reads, then fragments), arithmetic 4 per MFMA, writes last");
  return 0;
}
