// rt_rng_init.hip -- rng_init_kernel <- random::InitRandomStates / CreateStates (RayTracer/Random.cu:10-52).
// A translation unit of its own: it is compiled with the default machine scheduler (the max-ILP strategy the
// trace kernel wants hoists all 80 LDS look-ups of a table product and spills at the 128-VGPR budget).
#include "rt_device_math.hpp"
#include "rt_kernels.hpp"

namespace rtk {

using rtd::Rng;

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------
// RNG state creation: curand_init(seed, subsequence = global pixel index, offset 0), i.e.
// v_p = J^p v_seed with J = T^(2^67) (160 x 160 over GF(2)), jump[k] = J^(2^k) as 160 columns of 8 words
// (5 used).  All powers of J commute, so for the 64 consecutive pixels p = A + lane of a wave
//     v_p = J^lane (J^A v_seed):
//   * J^A v_seed is wave-uniform and computed COOPERATIVELY: per set bit k of A one matrix-vector product in
//     which lane l XORs the (at most three) columns l, l+64, l+128 whose bits are set, then a 6-step XOR
//     butterfly over the wave -- ~80 instructions per set bit for the whole wave instead of 1600 per lane;
//   * J^lane is per lane: the six matrices J^(2^m), m < 6, applied in lockstep through 4-bit window tables in
//     LDS (rt_rng_host.hpp: 40 look-ups of 20 bytes per product instead of 160 masked column XORs; the 16
//     entries of a window are 256 contiguous bytes, so the b128 reads are bank-conflict free).
// A block stages the 75 KiB of tables once and walks over the frame (grid-stride over waves); the stride is a
// power of two, so a wave steps from one chunk to its next with ONE cooperative product by J^stride whose
// columns it keeps in registers -- the full J^A product (a chain of ~10 dependent products fed from L2) is
// paid once per wave only.
// 1080p: 1.61 ms -> see profiles/ (the states are bit-identical: same GF(2) products, re-associated).
// ------------------------------------------------------------------------------------
constexpr uint32_t kRngWinEntries = 6u * 40u * 16u;
constexpr uint32_t kRngInitThreads = 512;

__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v ^= static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), off, 64));
  return v;
}

__global__ __launch_bounds__(kRngInitThreads, 4) void rng_init_kernel(uint32_t* __restrict__ rng, uint32_t npix,
                                                                    uint32_t p0, Rng seeded,
                                                                    const uint32_t* __restrict__ jump,
                                                                    const uint32_t* __restrict__ win) {
  extern __shared__ uint4 s_win[];                                   // kRngWinEntries x words 0..3, then x word 4
  uint32_t* const s_w4 = reinterpret_cast<uint32_t*>(s_win + kRngWinEntries);
  for (uint32_t i = threadIdx.x; i < kRngWinEntries; i += kRngInitThreads) {
    s_win[i] = reinterpret_cast<const uint4*>(win)[i];
    s_w4[i] = win[kRngWinEntries * 4u + i];
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t waves_per_block = kRngInitThreads / 64u;
  const uint32_t wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
  const uint32_t stride = gridDim.x * kRngInitThreads;               // pixels between a wave's chunks: a power of two (launcher)
  // one cooperative product u <- M u with M = 160 columns of 8 words: lane l takes the columns l, l+64, l+128
  auto own_word = [&](const uint32_t (&u)[5], uint32_t part) {
    return part == 0u ? (lane < 32u ? u[0] : u[1]) : part == 1u ? (lane < 32u ? u[2] : u[3]) : u[4];
  };
  // the columns of J^stride this lane owns stay in registers: the step from one chunk of the wave to its next
  uint32_t sc[3][5];
  {
    const uint32_t* __restrict__ M = jump + static_cast<size_t>(__builtin_ctz(stride)) * 160u * 8u;
#pragma unroll
    for (uint32_t part = 0; part < 3u; ++part) {
      const uint32_t bit = lane + part * 64u < 160u ? lane + part * 64u : 0u;
      const uint4 c = *reinterpret_cast<const uint4*>(M + bit * 8u);
      sc[part][0] = c.x; sc[part][1] = c.y; sc[part][2] = c.z; sc[part][3] = c.w; sc[part][4] = M[bit * 8u + 4u];
    }
  }
  uint32_t u[5] = {seeded.v0, seeded.v1, seeded.v2, seeded.v3, seeded.v4};   // wave-uniform, every lane holds it
  bool first = true;
  for (uint32_t base = wave0 * 64u; base < npix; base += stride) {   // wave-uniform
    if (first) {
      // ---- u = J^A v_seed for the wave's first chunk, A = subsequence of its first pixel ----
      const uint32_t A = p0 + base;
      for (uint32_t k = 0; k < 32u; ++k) {
        if (!((A >> k) & 1u)) continue;                              // wave-uniform
        const uint32_t* __restrict__ M = jump + static_cast<size_t>(k) * 160u * 8u;
        uint32_t r[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t part = 0; part < 3u; ++part) {
          const uint32_t bit = lane + part * 64u;                    // the state bit this lane owns in this part
          if (bit < 160u) {
            const uint32_t m = 0u - ((own_word(u, part) >> (lane & 31u)) & 1u);
            const uint4 c = *reinterpret_cast<const uint4*>(M + bit * 8u);
            const uint32_t c4 = M[bit * 8u + 4u];
            r[0] ^= c.x & m; r[1] ^= c.y & m; r[2] ^= c.z & m; r[3] ^= c.w & m; r[4] ^= c4 & m;
          }
        }
#pragma unroll
        for (int w = 0; w < 5; ++w) u[w] = wave_xor(r[w]);
      }
      first = false;
    } else {
      // ---- next chunk of this wave: u <- J^stride u, columns from registers ----
      uint32_t r[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
      for (uint32_t part = 0; part < 3u; ++part) {
        const bool owns = lane + part * 64u < 160u;
        const uint32_t m = owns ? 0u - ((own_word(u, part) >> (lane & 31u)) & 1u) : 0u;
#pragma unroll
        for (int w = 0; w < 5; ++w) r[w] ^= sc[part][w] & m;
      }
#pragma unroll
      for (int w = 0; w < 5; ++w) u[w] = wave_xor(r[w]);
    }
    // ---- v = J^lane u through the window tables: six products in lockstep, kept where the lane's bit is set ----
    uint32_t v[5] = {u[0], u[1], u[2], u[3], u[4]};
    for (uint32_t m = 0; m < 6u; ++m) {
      const uint4* __restrict__ T = s_win + m * 640u;
      const uint32_t* __restrict__ T4 = s_w4 + m * 640u;
      uint32_t r0 = 0u, r1 = 0u, r2 = 0u, r3 = 0u, r4 = 0u;
#pragma unroll 1
      for (uint32_t w = 0; w < 5u; ++w) {                            // the eight nibbles of one state word at a time
        const uint32_t vw = w == 0u ? v[0] : w == 1u ? v[1] : w == 2u ? v[2] : w == 3u ? v[3] : v[4];   // (no indexed register file)
        const uint4* __restrict__ Tw = T + w * 128u;
        const uint32_t* __restrict__ T4w = T4 + w * 128u;
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) {
          const uint32_t n = (vw >> (q * 4u)) & 15u;
          const uint4 e = Tw[q * 16u + n];
          r0 ^= e.x; r1 ^= e.y; r2 ^= e.z; r3 ^= e.w;
          r4 ^= T4w[q * 16u + n];
        }
      }
      if ((lane >> m) & 1u) { v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4; }
    }
    const uint32_t i = base + lane;
    if (i < npix) {
      rng[0 * static_cast<size_t>(npix) + i] = seeded.d;
      rng[1 * static_cast<size_t>(npix) + i] = v[0];
      rng[2 * static_cast<size_t>(npix) + i] = v[1];
      rng[3 * static_cast<size_t>(npix) + i] = v[2];
      rng[4 * static_cast<size_t>(npix) + i] = v[3];
      rng[5 * static_cast<size_t>(npix) + i] = v[4];
    }
  }
}

hipError_t launch_rng_init(uint32_t* rng, uint32_t npix, uint32_t p0, const uint32_t seeded[6],
                           const uint32_t* jump, const uint32_t* win, hipStream_t st) {
  Rng s = {seeded[0], seeded[1], seeded[2], seeded[3], seeded[4], seeded[5]};
  if (npix == 0) return hipSuccess;
  // two 512-thread blocks fit a CU's LDS (75 KiB of tables each, 8 waves per block: no register cap); small frames get one block per 512 pixels
  // a power of two, so that a wave's next chunk is ONE fixed jump J^(blocks * 512) away
  const uint32_t want = cdiv(npix, kRngInitThreads);
  uint32_t blocks = 1u;
  while (blocks < 512u && blocks * 2u <= want) blocks *= 2u;
  hipLaunchKernelGGL(rng_init_kernel, dim3(blocks), dim3(kRngInitThreads), kRngWinEntries * 20u, st, rng, npix, p0, s, jump, win);
  return hipGetLastError();
}

}  // namespace rtk
