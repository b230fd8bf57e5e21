// Second translation unit of liblm_engine.so: k_step compiled for two wavefronts per SIMD (k_step_w2) and its launcher.  The source is
// lm_engine.hip itself with LM_WAVES2 (<= 256 registers, <= 20 KB of LDS per wavefront; see the comment at STASH_SLOTS there) up to the end of
// k_step; lm_step dispatches it beyond 32 768 envs on locomotion engines (+ 5-8 % from 36 864 envs, + 9 % at 65 536, + 14 % at
// 131 072, + 18 % at 262 144; slower below 24 576).  A separate unit so that the
// one-wavefront kernels of lm_engine.hip keep their register allocation and code layout.
#ifndef LM_WAVES2
#define LM_WAVES2 1
#endif
#define LM_W2_UNIT 1
#undef LM_STAMPS              // diagnostic switches of the whole-library builds do not apply to this unit (their device globals live in lm_engine.hip)
#undef LM_COUNT_PASS2
#include "lm_engine.hip"
