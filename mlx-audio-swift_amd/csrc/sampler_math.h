// sampler_math.h - the arithmetic "mis-sampler-v1" fixes bit for bit (oracle/sampler.py), shared by every device sampler
// (lm_sampler.hip, token_engine.hip): the deterministic float32 exp and the fixed-point scale of the probability masses.
#pragma once
#include "common.h"

#define E_SCALE 1099511627776.0f        // 2^40: E_i = trunc(e_i * 2^40)

// exp(y) for y <= 0 as a fixed sequence of IEEE float32 multiplies and adds (no contraction): 2^f on [0, 1) by a degree-6 polynomial,
// scaled by 2^n; 0 below 2^-60
__device__ __forceinline__ float det_exp_dev(float y) {
#pragma clang fp contract(off)
    const float LOG2E = 1.4426950408889634f;
    float t = y * LOG2E;
    float n = floorf(t);
    float f = t - n;
    float p = 0.00015403530393381608f;
    p = p * f; p = p + 0.0013333558146428443f;
    p = p * f; p = p + 0.009618129107628477f;
    p = p * f; p = p + 0.05550410866482158f;
    p = p * f; p = p + 0.2402265069591007f;
    p = p * f; p = p + 0.6931471805599453f;
    p = p * f; p = p + 1.0f;
    int ni = (int)fmaxf(n, -64.0f);
    float r = p * ldexpf(1.0f, ni);
    return (n < -60.0f) ? 0.0f : r;
}
