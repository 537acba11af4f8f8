// rt_rng_host.hpp -- host side of the XORWOW state creation.
//
// curand_init(seed, subsequence, 0) (call site RayTracer/Random.cu:22-27) = seed scramble
// + a jump of subsequence * 2^67 steps of the xorshift part.  The xorshift step is linear
// over GF(2) on 160 bits, so the jump is the matrix J = T^(2^67); we build J by 67
// squarings of the one-step matrix T and then J^(2^k), k = 0..31, which the device kernel
// multiplies together according to the bits of the pixel index.  Nothing is copied from
// cuRAND's precomputed tables (they are not available here); the matrices follow from the
// recurrence alone.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace rth {

struct Gf2Mat { uint32_t col[160][5]; };   // column i = image of basis vector e_i

inline void xorshift_step(uint32_t v[5]) {
  const uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

inline void mat_vec(const Gf2Mat& A, const uint32_t x[5], uint32_t out[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 160; ++i)
    if ((x[i >> 5] >> (i & 31)) & 1u)
      for (int w = 0; w < 5; ++w) r[w] ^= A.col[i][w];
  memcpy(out, r, sizeof r);
}

inline void mat_mul(const Gf2Mat& A, const Gf2Mat& B, Gf2Mat& out) {   // out = A * B
  Gf2Mat tmp;
  for (int i = 0; i < 160; ++i) mat_vec(A, B.col[i], tmp.col[i]);
  out = tmp;
}

// jump[k] = (T^(2^67))^(2^k), k = 0..31, as 160 columns padded to 8 words (device layout)
inline std::vector<uint32_t> build_jump_table() {
  Gf2Mat m;
  for (int i = 0; i < 160; ++i) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[i >> 5] = 1u << (i & 31);
    xorshift_step(v);
    memcpy(m.col[i], v, sizeof v);
  }
  for (int i = 0; i < 67; ++i) mat_mul(m, m, m);
  std::vector<uint32_t> table(32u * 160u * 8u, 0u);
  for (int k = 0; k < 32; ++k) {
    for (int i = 0; i < 160; ++i)
      memcpy(&table[(static_cast<size_t>(k) * 160u + i) * 8u], m.col[i], 5 * sizeof(uint32_t));
    mat_mul(m, m, m);
  }
  return table;
}

// seed scramble of curand_init: state = {d, v0..v4}
inline void seed_state(uint64_t seed, uint32_t s[6]) {
  const uint32_t lo = static_cast<uint32_t>(seed) ^ 0xaad26b49u;
  const uint32_t hi = static_cast<uint32_t>(seed >> 32) ^ 0xf7dcefddu;
  const uint32_t t0 = 1099087573u * lo;
  const uint32_t t1 = 2591861531u * hi;
  s[0] = 6615241u + t1 + t0;
  s[1] = 123456789u + t0;
  s[2] = 362436069u ^ t0;
  s[3] = 521288629u + t1;
  s[4] = 88675123u ^ t1;
  s[5] = 5783321u + t0;
}

// host evaluation of curand_init(seed, subsequence, 0) from the same table (debug/tests)
inline void init_state(const std::vector<uint32_t>& table, uint64_t seed, uint64_t subsequence,
                       uint32_t s[6]) {
  seed_state(seed, s);
  for (int k = 0; k < 32; ++k) {
    if (!((subsequence >> k) & 1u)) continue;
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 160; ++i)
      if ((s[1 + (i >> 5)] >> (i & 31)) & 1u)
        for (int w = 0; w < 5; ++w) r[w] ^= table[(static_cast<size_t>(k) * 160u + i) * 8u + w];
    memcpy(s + 1, r, sizeof r);
  }
}

}  // namespace rth
