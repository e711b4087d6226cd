// tools/verify_exp_nonpos.c -- dexpf_nonpos(x) == dexpf(x), bit for bit, for every float x <= 0 (-0, denormals, -inf included), and
// ldexpf(p, n) == the specification's two-step scaling for every x of dexpf's arithmetic range (the device code uses v_ldexp_f32).
// Host restatement of the two device functions of parakeet.cpp_amd/csrc/pk_devmath.h (same constants, fmaf = correctly rounded fma,
// build with -ffp-contract=off).  usage: verify_exp_nonpos [stride]   (stride 1 = all 2 139 095 042 values, ~80 s on one core)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float dexpf_ref(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -87.33654022216797f) return 0.0f;
    const float t = fmaf(x, 1.44269502162933349609375f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float e = 0x1.6d4332p-10f;
    e = fmaf(e, r, 0x1.120b74p-7f);
    e = fmaf(e, r, 0x1.5554e8p-5f);
    e = fmaf(e, r, 0x1.5554dcp-3f);
    e = fmaf(e, r, 0.5f);
    const float q = fmaf(r * r, e, r);
    const float p = q + 1.0f;
    const int ni = (int)n;
    const int n1 = ni >> 1;
    const int n2 = ni - n1;
    const float s1 = asf((uint32_t)(n1 + 127) << 23);
    const float s2 = asf((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}
static float dexpf_np(float x) {
    if (x < -87.33654022216797f) return 0.0f;
    const float t = fmaf(x, 1.44269502162933349609375f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float e = 0x1.6d4332p-10f;
    e = fmaf(e, r, 0x1.120b74p-7f);
    e = fmaf(e, r, 0x1.5554e8p-5f);
    e = fmaf(e, r, 0x1.5554dcp-3f);
    e = fmaf(e, r, 0.5f);
    const float q = fmaf(r * r, e, r);
    const float p = q + 1.0f;
    const int ni = (int)n;
    return p * asf((uint32_t)(ni + 127) << 23);
}
int main(int argc, char **argv) {
    unsigned long long bad = 0, cnt = 0;
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    for (uint64_t u = 0x80000000ull; u <= 0xff800000ull; u += (u >= 0xff800000ull - stride ? 1 : stride)) {     // -0 .. -inf
        const float x = asf((uint32_t)u);
        const uint32_t a = asu(dexpf_ref(x)), b = asu(dexpf_np(x));
        ++cnt;
        if (a != b) { if (bad < 5) printf("x=%a ref=%a np=%a\n", x, asf(a), asf(b)); ++bad; }
    }
    // second property: dexpf's scaling p 2^n as one ldexpf == the specification's two multiplications, for every x of the arithmetic range
    for (uint64_t u = 0; u <= 0xffffffffull; u += stride) {
        const float x = asf((uint32_t)u);
        if (!(x >= -87.33654022216797f && x <= 88.72283935546875f)) continue;
        const float t = fmaf(x, 1.44269502162933349609375f, 12582912.0f);
        const float n = t - 12582912.0f;
        float r = fmaf(n, -0.693145751953125f, x);
        r = fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
        float e = 0x1.6d4332p-10f;
        e = fmaf(e, r, 0x1.120b74p-7f);
        e = fmaf(e, r, 0x1.5554e8p-5f);
        e = fmaf(e, r, 0x1.5554dcp-3f);
        e = fmaf(e, r, 0.5f);
        const float p = fmaf(r * r, e, r) + 1.0f;
        const int ni = (int)n, n1 = ni >> 1, n2 = ni - n1;
        const float two = (p * asf((uint32_t)(n1 + 127) << 23)) * asf((uint32_t)(n2 + 127) << 23);
        ++cnt;
        if (asu(two) != asu(ldexpf(p, ni))) { if (bad < 5) printf("ldexp: x=%a\n", x); ++bad; }
    }
    const float z = 0.0f;
    if (asu(dexpf_ref(z)) != asu(dexpf_np(z))) ++bad;
    printf("checked %llu values, %llu mismatches\n", cnt + 1, bad);
    return bad != 0;
}
