// lm_rng.h -- the counter-based random streams shared by the step engine (goal sampling, domain randomisation) and the policy
// kernels (gaussian action sampling).  No generator state: a draw is a pure function of (seed, stream, env, key, index), so kernels
// can be captured into graphs, replayed and checkpointed, and the CPU oracle (oracle/lm_oracle.c: mix32, lmo_hash_uniform3,
// lmo_dr_sample) reproduces every bit of the integer part.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LM_RNG_DEV __device__ __forceinline__

LM_RNG_DEV uint32_t lm_mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }   // lowbias32 finaliser

// hash of (seed, stream, env, key): one per (env, quantity, episode / step)
LM_RNG_DEV uint32_t lm_rng_base(uint32_t seed, uint32_t stream, uint32_t env, uint32_t key) {
  return lm_mix32(seed ^ lm_mix32(env * 0x9E3779B9U + 0x7F4A7C15U) ^ lm_mix32(key * 0x85EBCA6BU + 0x165667B1U) ^ lm_mix32(stream * 0x27D4EB2FU + 0x632BE5ABU));
}
// the pair of uniforms shared by components 2p and 2p+1: u1 in (0, 1], u2 in [0, 1)
LM_RNG_DEV void lm_rng_pair(uint32_t base, uint32_t pair, float* u1, float* u2) {
  const uint32_t r1 = lm_mix32(base + (2U * pair + 1U) * 0xC2B2AE35U), r2 = lm_mix32(base + (2U * pair + 2U) * 0xC2B2AE35U);
  *u1 = ((float)(r1 >> 8) + 1.0f) * (1.0f / 16777216.0f); *u2 = (float)(r2 >> 8) * (1.0f / 16777216.0f);
}
#define LM_RNG_STREAM_ACTION_SAMPLING 9U      // streams 0..8 are the domain-randomisation channels (include/lm_engine.h)
