/* exact_div_check.c -- exhaustive / mass check, on the CPU, that the division the ray-finished phase uses on ordinary operands,
 *     y = RN(1/b);  q = RN(a*y);  q' = RN(q + RN(a - b*q) * y)          (Markstein's correction step, two FMAs)
 * is the correctly rounded quotient RN(a/b) for the operand families it is used on (csrc/vxrt_device.hpp: div_rn):
 *   1. the tonemap c / (c + 1), every binary32 c with 2^-100 <= c <= 2^100 (and c = 0);
 *   2. pixel coordinates x / W, y / H: all integers 0 <= x <= 65535, 1 <= W <= 65535;
 *   3. the occlusion mean s / n: s a multiple of 0.5 up to n, n <= 4096;
 *   4. random pairs of ordinary size (exponents in [-60, 60]).
 * y = RN(1/b) is the hardware's v_rcp_f32 + one Newton step on the GPU, verified equal to the IEEE reciprocal on every
 * binary32 of exponent [-100, 100] by tools/ubench/rcp_check.hip; here it is the IEEE reciprocal itself.
 * build: gcc -O2 -mfma -ffp-contract=off -fopenmp exact_div_check.c -lm     usage: exact_div_check [stride=1] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float div_rn(float a, float b)
{
    const float y = 1.0f / b;
    const float q = a * y;
    return fmaf(fmaf(-b, q, a), y, q);
}
int main(int argc, char **argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    unsigned long long bad1 = 0, bad2 = 0, bad3 = 0, bad4 = 0, n1 = 0, n2 = 0, n4 = 0;
    const uint32_t lo = (127u - 100u) << 23, hi = (127u + 100u) << 23;
#pragma omp parallel for reduction(+ : bad1, n1) schedule(static)
    for (uint32_t b = lo; b <= hi; b += stride) {
        const float c = from_bits(b), d = c + 1.0f;
        n1 += 1;
        if (bits_of(div_rn(c, d)) != bits_of(c / d))
            bad1 += 1;
    }
    if (bits_of(div_rn(0.0f, 1.0f)) != bits_of(0.0f / 1.0f))
        bad1 += 1;
#pragma omp parallel for reduction(+ : bad2, n2) schedule(static)
    for (uint32_t W = 1; W <= 65535u; W += (stride > 1 ? 7u : 1u)) {
        const float wf = (float)(int)W;
        for (uint32_t x = 0; x <= 65535u; ++x) {
            const float xf = (float)(int)x;
            n2 += 1;
            if (bits_of(div_rn(xf, wf)) != bits_of(xf / wf))
                bad2 += 1;
        }
    }
    for (int n = 1; n <= 4096; ++n)
        for (int s2 = 0; s2 <= 2 * n; ++s2) {
            const float s = 0.5f * (float)s2, nf = (float)n;
            if (bits_of(div_rn(s, nf)) != bits_of(s / nf))
                bad3 += 1;
        }
#pragma omp parallel for reduction(+ : bad4, n4) schedule(static)
    for (int t = 0; t < 64; ++t) {
        uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
        for (uint32_t i = 0; i < (stride > 1 ? 2000000u : 16000000u); ++i) {
            z ^= z << 13; z ^= z >> 7; z ^= z << 17;
            const uint32_t ea = 127u - 60u + (uint32_t)((z >> 8) % 121u), eb = 127u - 60u + (uint32_t)((z >> 20) % 121u);
            const float a = from_bits(((uint32_t)(z >> 33) & 0x807FFFFFu) | (ea << 23)), b = from_bits(((uint32_t)z & 0x807FFFFFu) | (eb << 23));
            n4 += 1;
            if (bits_of(div_rn(a, b)) != bits_of(a / b))
                bad4 += 1;
        }
    }
    printf("tonemap c/(c+1): %llu of %llu differ; x/W: %llu of %llu; s/n: %llu; random pairs: %llu of %llu\n", bad1, n1, bad2, n2, bad3, bad4, n4);
    return (bad1 | bad2 | bad3 | bad4) != 0;
}
