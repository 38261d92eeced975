// mfma_quadform.hip -- ONE experiment: does the forward blend's visit get cheaper when the quadratic form
// log2(G) = A' dx^2 + B' dx dy + C' dy^2 of its alpha test is evaluated on the MATRIX pipe instead of with five vector instructions?
//
// The blend's layout is lane = pixel, loop over the tile's instances (the compositing recurrence is sequential per pixel), so the only
// MFMA whose result lands where the visit needs it is v_mfma_f32_4x4x1_16b_f32: sixteen 4x4 blocks, block b = lanes 4b..4b+3, lane
// 4b + j holds column j of D for rows i = 0..3 in four registers.  With the A operand (four instances' coefficient k) broadcast from
// block 0 to all blocks (cbsz = 4) and the B operand = the lane's own pixel monomial k, D[i] of lane l = sum_k coef_k[instance i] *
// mono_k[pixel l]: six instructions (monomials 1, u, v, u^2, uv, v^2 in sub-tile-centred coordinates) give FOUR instances' log2(G)
// for all 64 pixels, in the lane-is-pixel layout, exact binary32 fused multiply-adds.  (The 16x16x4 / 32x32x2 shapes spread one
// pixel's values over several lanes: they would need a transpose through LDS, 16 kB per wave.)
//
// This file times the visit loop of k_blend_fwd_w6 (alpha test, hit ballot, blend body -- the kernel's own statements) on synthetic
// staged batches in both forms, at the real kernel's launch shape (single-wave workgroups, eight per SIMD, 5 kB of LDS each):
//   V  the kernel's form: per visit two LDS reads (x, y, A', B' | C', opacity), dx, dy, two multiplies, two fmas;
//   M  per group of four instances six LDS reads by lanes 0..3 + six MFMAs, per visit one LDS read (opacity).
// Measured on MI355X (profiles/r06_mfma_quadform.txt): M / V = 1.01 and 0.99 at 6 and 12 batches per wave -- the same time.  The visit is
// not bound by the five vector instructions the matrix pipe can take over.
// Both forms visit every staged instance (no reach bits) and must produce the same pixel sums to 1e-5.
// Build + run: tools/microbench/run_mfma_quadform.sh -> one line per form: us per launch, ns per (wave, instance) visit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <cmath>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr float kAlphaMax = 0.99f, kAlphaMin = 1.f / 255.f, kTStop = 1e-4f, kL2E = 1.4426950408889634f;

struct Rec { float x, y, a, b, c, op, depth, r, g, bl, pad0, pad1; };   // a staged instance: mean relative to the sub-tile's first pixel, conic, opacity, colour

template <int FORM>
__global__ __launch_bounds__(64) void k_visit(const Rec* __restrict__ recs, int nrec, int batches, float* __restrict__ out)
{
    __shared__ float4 s_ab[2][64];     // (x, y, A', B') | (C', opacity, depth, r)
    __shared__ float2 s_c[64];         // (g, b)
    __shared__ float s_coef[6][64];    // FORM 1: the six coefficients of log2(G) per staged instance
    const int lane = threadIdx.x;
    const float pxf = (float)(lane & 7), pyf = (float)(lane >> 3);
    const float u = pxf - 3.5f, v = pyf - 3.5f;                     // sub-tile-centred pixel coordinates: |u|, |v| <= 3.5
    const float mono[6] = {1.f, u, v, u * u, u * v, v * v};
    float Tr = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dd = 0.f, Aa = 0.f;
    uint32_t last = 0u;
    size_t base = ((size_t)blockIdx.x * 131u) % (size_t)nrec;
    for (int b = 0; b < batches; b++) {
        const Rec rc = recs[(base + (size_t)b * 64 + lane) % (size_t)nrec];
        const float A = rc.a * (-0.5f * kL2E), B = rc.b * (-kL2E), C = rc.c * (-0.5f * kL2E);
        s_ab[0][lane] = make_float4(rc.x, rc.y, A, B);
        s_ab[1][lane] = make_float4(C, rc.op, rc.depth, rc.r);
        s_c[lane] = make_float2(rc.g, rc.bl);
        if (FORM == 1) {       // log2 G = A (mx - u)^2 + B (mx - u)(my - v) + C (my - v)^2 with (mx, my) = the mean relative to the sub-tile's centre
            const float mx = rc.x - 3.5f, my = rc.y - 3.5f;
            s_coef[0][lane] = fmaf(C * my, my, fmaf(B, my, A * mx) * mx);
            s_coef[1][lane] = -(2.f * A * mx + B * my);
            s_coef[2][lane] = -(2.f * C * my + B * mx);
            s_coef[3][lane] = A; s_coef[4][lane] = B; s_coef[5][lane] = C;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto blend = [&](int j, float p2, float alpha) {          // k_blend_fwd_w6's visit body
            const bool hit = !(p2 > 0.f) && !(alpha < kAlphaMin);
            if ((__builtin_amdgcn_ballot_w64(!(p2 > 0.f)) & __builtin_amdgcn_ballot_w64(!(alpha < kAlphaMin))) == 0ull) return;
            const float4 Bq = s_ab[1][j];
            const float2 Cq = s_c[j];
            const float am = hit ? alpha : 0.f;
            const float test_T = Tr * (1.f - am);
            const bool pass = !(test_T < kTStop);
            const float asel = pass ? am : 0.f;
            const float w = asel * Tr;
            C0 = fmaf(Bq.w, w, C0); C1 = fmaf(Cq.x, w, C1); C2 = fmaf(Cq.y, w, C2);
            Dd = fmaf(Bq.z, w, Dd); Aa = fmaf(Tr, asel, Aa);
            Tr = pass ? test_T : -fabsf(Tr);
            last = (pass && hit) ? (uint32_t)(b * 64 + j + 1) : last;
        };
        if (FORM == 0) {
            for (int j = 0; j < 64; j++) {
                const float4 Aq = s_ab[0][j];
                const float2 Bq = *reinterpret_cast<const float2*>(&s_ab[1][j]);
                const float dx = Aq.x - pxf, dy = Aq.y - pyf;
                const float p2 = fmaf(Bq.x * dy, dy, fmaf(Aq.w, dy, Aq.z * dx) * dx);
                const float alpha = fminf(kAlphaMax, Bq.y * __builtin_amdgcn_exp2f(p2));
                blend(j, p2, alpha);
            }
        } else {
#pragma unroll 1
            for (int g = 0; g < 16; g++) {
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                const int src = 4 * g + (lane & 3);               // (only block 0 -- lanes 0..3 -- is read by the broadcast)
#pragma unroll
                for (int k = 0; k < 6; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(s_coef[k][src], mono[k], acc, 4, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = 4 * g + i;
                    const float p2 = acc[i];
                    const float alpha = fminf(kAlphaMax, s_ab[1][j].y * __builtin_amdgcn_exp2f(p2));
                    blend(j, p2, alpha);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    out[(size_t)blockIdx.x * 64 + lane] = C0 + C1 + C2 + Dd + Aa + fabsf(Tr) + (float)last * 1e-6f;
}

int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 8960, batches = argc > 2 ? atoi(argv[2]) : 6, reps = 20, nrec = 1 << 16;
    std::vector<Rec> h(nrec);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.f / 16777216.f); };
    for (auto& r : h) {      // splats of ~3 px sigma around the 8x8 block, a third of them too far to reach it: the mix a tile's list holds
        r.x = -6.f + 20.f * rnd(); r.y = -6.f + 20.f * rnd();
        const float sx = 1.5f + 3.f * rnd(), sy = 1.5f + 3.f * rnd(), rho = 0.6f * (rnd() - 0.5f);
        const float det = sx * sx * sy * sy * (1.f - rho * rho);
        r.a = sy * sy / det; r.c = sx * sx / det; r.b = -2.f * rho * sx * sy / det;     // conic of the 2x2 covariance; B as the kernels store it (x2)
        r.op = 0.02f + 0.5f * rnd() * rnd(); r.depth = 1.f + 9.f * rnd(); r.r = rnd(); r.g = rnd(); r.bl = rnd(); r.pad0 = r.pad1 = 0.f;
    }
    Rec* d_rec; float *d_out0, *d_out1;
    HIPCHECK(hipMalloc(&d_rec, nrec * sizeof(Rec)));
    HIPCHECK(hipMalloc(&d_out0, (size_t)waves * 64 * 4));
    HIPCHECK(hipMalloc(&d_out1, (size_t)waves * 64 * 4));
    HIPCHECK(hipMemcpy(d_rec, h.data(), nrec * sizeof(Rec), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    float us[2] = {0, 0};
    for (int form = 0; form < 2; form++) {
        for (int r = 0; r < reps + 3; r++) {
            if (r == 3) HIPCHECK(hipEventRecord(e0));
            if (form == 0) hipLaunchKernelGGL(k_visit<0>, dim3(waves), dim3(64), 0, 0, d_rec, nrec, batches, d_out0);
            else hipLaunchKernelGGL(k_visit<1>, dim3(waves), dim3(64), 0, 0, d_rec, nrec, batches, d_out1);
        }
        HIPCHECK(hipEventRecord(e1));
        HIPCHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        us[form] = 1e3f * ms / reps;
    }
    std::vector<float> o0((size_t)waves * 64), o1(o0.size());
    HIPCHECK(hipMemcpy(o0.data(), d_out0, o0.size() * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(o1.data(), d_out1, o1.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    size_t bad = 0;
    for (size_t i = 0; i < o0.size(); i++) { const double d = fabs((double)o0[i] - o1[i]); md = d > md ? d : md; mx = fabs(o0[i]) > mx ? fabs(o0[i]) : mx; bad += d > 1e-4; }
    const double visits = (double)waves * batches * 64;
    printf("waves %d, batches of 64 instances per wave %d, (wave, instance) visits per launch %.0f\n", waves, batches, visits);
    // (per visit AND SIMD: 1 024 SIMDs work side by side -- the forward blend's own figure is 43-48 ns, tools/k6_lone_wave.py)
    printf("V  vector form (the kernel's)            %8.1f us per launch  %6.1f ns per visit and SIMD\n", us[0], 1e3 * us[0] * 1024.0 / visits);
    printf("M  v_mfma_f32_4x4x1_16b_f32 quadratic form %6.1f us per launch  %6.1f ns per visit and SIMD   (M / V = %.3f)\n", us[1], 1e3 * us[1] * 1024.0 / visits, us[1] / us[0]);
    printf("largest difference of a pixel's sums between the forms %.3g (largest sum %.3g); pixels differing by more than 1e-4: %zu of %zu\n", md, mx, bad, o0.size());
    return 0;
}
