/*
 * s3o_encoders.c — CPU ORACLE for shencoder, freqencoder and ffmlp.
 *
 * TEST INFRASTRUCTURE ONLY (see s3o_common.h).
 *
 * Spherical harmonics (shencoder/src/shencoder.cu:27-382).  The reference
 * hard-codes, for degree <= 8, the 64 real SH basis functions and their 3x64
 * Jacobian as expanded fp32 polynomials of the form
 *     Y_l^{+m} = (-1)^m sqrt2 N_l^m T_l^m(z) Re (x+iy)^m      (m > 0)
 *     Y_l^{-m} = (-1)^m sqrt2 N_l^m T_l^m(z) Im (x+iy)^m
 *     Y_l^0    = N_l^0 T_l^0(z)
 * with T_l^m = d^m P_l / dz^m (a polynomial in z alone, i.e. the unit-sphere
 * form — e.g. outputs[6] = c*(3 z^2 - 1), :60) and output slot l*l + l + m.
 * The oracle evaluates exactly this family through its defining recurrences in
 * DOUBLE precision and rounds once to fp32, so it is the mathematical value
 * the reference's fp32 polynomials approximate (agreement ~1e-6, FP tolerance
 * in the tests).  PINNED: against golden vectors obtained by evaluating the
 * reference's own expressions (oracle/gen_golden.py, SH section) and against
 * testing/test_shencoder.py's closed-form torch encoder for degree <= 5.
 *
 * freq_encode follows freqencoder/src/freqencoder.cu:30-94 literally.
 * ffmlp follows the layouts of ffmlp/src/ffmlp.cu:631-634, 742-748 and the
 * activations of ffmlp/src/utils.h:424-582; its numerics are the dense math
 * Y = act(X W^T) with fp16 storage between layers and fp32 accumulation (the
 * reference accumulates in fp16 inside WMMA, which no other hardware
 * reproduces — pinned against the bias-free torch MLP twin of
 * testing/test_ffmlp.py:11-43 with an fp16 tolerance).
 */
#include "s3o_common.h"
#include <stdlib.h>

#define SH_MAXDEG 8

/* inputs [B,3], outputs [B,deg^2], dy_dx [B,3,deg^2] or NULL */
S3O_API void s3o_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D,
                                   uint32_t degree, float* dy_dx) {
    const uint32_t C2 = degree * degree;
    /* normalisation table K[l][m] = N_l^m * (m ? (-1)^m sqrt2 : 1) */
    double K[SH_MAXDEG][SH_MAXDEG];
    for (uint32_t l = 0; l < degree; l++)
        for (uint32_t m = 0; m <= l; m++) {
            double ratio = 1.0; /* (l-m)! / (l+m)! */
            for (uint32_t k = l - m + 1; k <= l + m; k++) ratio /= (double)k;
            double n = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * ratio);
            if (m) n *= ((m & 1) ? -1.0 : 1.0) * M_SQRT2;
            K[l][m] = n;
        }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        const double x = inputs[b * D], y = inputs[b * D + 1], z = inputs[b * D + 2];
        double c[SH_MAXDEG + 1], s[SH_MAXDEG + 1]; /* Re/Im (x+iy)^m */
        c[0] = 1; s[0] = 0;
        for (uint32_t m = 1; m <= degree; m++) { c[m] = x * c[m - 1] - y * s[m - 1]; s[m] = x * s[m - 1] + y * c[m - 1]; }
        /* T[l][m] = d^m P_l/dz^m, m up to l (T[l][l+1] = 0) */
        double T[SH_MAXDEG][SH_MAXDEG + 2];
        memset(T, 0, sizeof(T));
        for (uint32_t m = 0; m < degree; m++) {
            double dfact = 1.0; /* (2m-1)!! */
            for (uint32_t k = 1; k <= m; k++) dfact *= (2.0 * k - 1.0);
            T[m][m] = dfact;
            if (m + 1 < degree) T[m + 1][m] = (2.0 * m + 1.0) * z * dfact;
            for (uint32_t l = m + 2; l < degree; l++)
                T[l][m] = ((2.0 * l - 1.0) * z * T[l - 1][m] - (double)(l + m - 1) * T[l - 2][m]) / (double)(l - m);
        }
        float* o = outputs + (size_t)b * C2;
        float* jx = dy_dx ? dy_dx + (size_t)b * 3 * C2 : NULL;
        float* jy = jx ? jx + C2 : NULL;
        float* jz = jx ? jx + 2 * C2 : NULL;
        for (uint32_t l = 0; l < degree; l++) {
            const uint32_t base = l * l + l;
            o[base] = (float)(K[l][0] * T[l][0]);
            if (jx) { jx[base] = 0; jy[base] = 0; jz[base] = (float)(K[l][0] * T[l][1]); }
            for (uint32_t m = 1; m <= l; m++) {
                const double kt = K[l][m] * T[l][m];
                o[base + m] = (float)(kt * c[m]);
                o[base - m] = (float)(kt * s[m]);
                if (jx) {
                    const double kz = K[l][m] * T[l][m + 1];
                    jx[base + m] = (float)(kt * m * c[m - 1]);
                    jx[base - m] = (float)(kt * m * s[m - 1]);
                    jy[base + m] = (float)(-kt * m * s[m - 1]);
                    jy[base - m] = (float)(kt * m * c[m - 1]);
                    jz[base + m] = (float)(kz * c[m]);
                    jz[base - m] = (float)(kz * s[m]);
                }
            }
        }
    }
}

/* shencoder.cu:358-382: grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch] */
S3O_API void s3o_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D,
                                    uint32_t degree, const float* dy_dx, float* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; t++) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float* g = grad + (size_t)b * C2;
        const float* j = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
        float acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(g[ch], j[ch], acc);
        grad_inputs[t] = acc;
    }
}

/* freqencoder.cu:30-58.  C = D + 2*D*deg. */
S3O_API void s3o_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg,
                                     uint32_t C, float* outputs) {
    (void)deg;
    const float half_pi = 3.141592653589793f / 2; /* PI()/2 in float, :54 */
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * C; t++) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (int64_t)b * C);
        const float* x = inputs + (size_t)b * D;
        if (c < D) { outputs[t] = x[c]; continue; }
        const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
        const float phase_shift = (float)(col % 2) * half_pi;
        outputs[t] = sinf(scalbnf(x[d], (int)freq) + phase_shift);
    }
}

/* freqencoder.cu:63-94 */
S3O_API void s3o_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D,
                                      uint32_t deg, uint32_t C, float* grad_inputs) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; t++) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float* g = grad + (size_t)b * C;
        const float* o = outputs + (size_t)b * C;
        float result = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; f++) {
            /* 2^f * (g_sin * cos - g_cos * sin); inner difference fused as nvcc would */
            result = fmaf(scalbnf(1.0f, (int)f), fmaf(g[d], o[D + d], -(g[D + d] * o[d])), result);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = result;
    }
}

/* ---- ffmlp ---- */
enum { ACT_RELU = 0, ACT_EXP = 1, ACT_SINE = 2, ACT_SIGMOID = 3, ACT_SQUAREPLUS = 4, ACT_SOFTPLUS = 5, ACT_NONE = 6 };
#define K_ACT 10.0f

static inline float act_fwd(uint32_t a, float x) { /* utils.h:424-474 */
    switch (a) {
        case ACT_RELU: return x > 0.0f ? x : 0.0f;
        case ACT_EXP: return expf(x);
        case ACT_SINE: return sinf(x);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case ACT_SQUAREPLUS: { float y = x * K_ACT; return 0.5f * (y + sqrtf(y * y + 4)) / K_ACT; }
        case ACT_SOFTPLUS: return logf(expf(x * K_ACT) + 1.0f) / K_ACT;
        default: return x;
    }
}
/* derivative from the stored POST-activation value, utils.h:534-582 */
static inline float act_bwd(uint32_t a, float g, float fwd) {
    switch (a) {
        case ACT_RELU: return fwd > 0.0f ? g : 0.0f;
        case ACT_EXP: return g * fwd;
        case ACT_SINE: return g; /* unsupported in the reference (returns without writing) */
        case ACT_SIGMOID: return g * (fwd * (1.0f - fwd));
        case ACT_SQUAREPLUS: { float y = fwd * K_ACT; return g * (y * y / (y * y + 1)); }
        case ACT_SOFTPLUS: return g * (1.0f - expf(-fwd * K_ACT));
        default: return g;
    }
}

/*
 * inputs [B,in] f16; weights f16 = [W,in] | (n-1) x [W,W] | [out,W] row-major
 * (ffmlp.cu:631-634); forward_buffer [n,B,W] f16 post-activation (may be NULL =
 * inference); outputs [B,out] f16.
 */
S3O_API void s3o_ffmlp_forward(const s3o_half* inputs, const s3o_half* weights, uint32_t B,
                               uint32_t in_dim, uint32_t out_dim, uint32_t W, uint32_t n_layers,
                               uint32_t activation, uint32_t out_activation,
                               s3o_half* forward_buffer, s3o_half* outputs) {
    const size_t nw = (size_t)W * in_dim + (size_t)W * W * (n_layers - 1) + (size_t)out_dim * W;
    float* wf = (float*)malloc(nw * sizeof(float));
    for (size_t i = 0; i < nw; i++) wf[i] = s3o_h2f(weights[i]);
#pragma omp parallel
    {
        float* cur = (float*)malloc(sizeof(float) * (W > in_dim ? W : in_dim));
        float* nxt = (float*)malloc(sizeof(float) * W);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            for (uint32_t i = 0; i < in_dim; i++) cur[i] = s3o_h2f(inputs[(size_t)b * in_dim + i]);
            const float* wl = wf;
            uint32_t k = in_dim;
            for (uint32_t l = 0; l < n_layers; l++) {
                for (uint32_t o = 0; o < W; o++) {
                    float acc = 0;
                    for (uint32_t i = 0; i < k; i++) acc = fmaf(cur[i], wl[(size_t)o * k + i], acc);
                    const s3o_half h = s3o_f2h(act_fwd(activation, s3o_h2f(s3o_f2h(acc))));
                    nxt[o] = s3o_h2f(h);
                    if (forward_buffer) forward_buffer[((size_t)l * B + b) * W + o] = h;
                }
                memcpy(cur, nxt, sizeof(float) * W);
                wl += (size_t)W * k;
                k = W;
            }
            for (uint32_t o = 0; o < out_dim; o++) {
                float acc = 0;
                for (uint32_t i = 0; i < W; i++) acc = fmaf(cur[i], wl[(size_t)o * W + i], acc);
                outputs[(size_t)b * out_dim + o] = s3o_f2h(act_fwd(out_activation, s3o_h2f(s3o_f2h(acc))));
            }
        }
        free(cur); free(nxt);
    }
    free(wf);
}

/*
 * grad [B,out] f16; backward_buffer [n,B,W] f16 (written: [k] = dL/d(pre-act of
 * hidden layer n-1-k)); grad_inputs [B,in] f16 or NULL; grad_weights f32 here
 * (test hook: the product returns f16, compared after rounding) — same flat
 * layout as weights.  ffmlp.cu:742-895.
 */
S3O_API void s3o_ffmlp_backward(const s3o_half* grad, const s3o_half* inputs, const s3o_half* weights,
                                const s3o_half* forward_buffer, uint32_t B, uint32_t in_dim,
                                uint32_t out_dim, uint32_t W, uint32_t n_layers, uint32_t activation,
                                s3o_half* backward_buffer, s3o_half* grad_inputs,
                                float* grad_weights) {
    const size_t n_first = (size_t)W * in_dim, n_hid = (size_t)W * W;
    const size_t nw = n_first + n_hid * (n_layers - 1) + (size_t)out_dim * W;
    float* wf = (float*)malloc(nw * sizeof(float));
    for (size_t i = 0; i < nw; i++) wf[i] = s3o_h2f(weights[i]);
    const float* w_last = wf + n_first + n_hid * (n_layers - 1);
    /* activation-gradient chain, per row */
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        /* through the last layer */
        for (uint32_t j = 0; j < W; j++) {
            float acc = 0;
            for (uint32_t o = 0; o < out_dim; o++)
                acc = fmaf(s3o_h2f(grad[(size_t)b * out_dim + o]), w_last[(size_t)o * W + j], acc);
            const float f = s3o_h2f(forward_buffer[((size_t)(n_layers - 1) * B + b) * W + j]);
            backward_buffer[((size_t)0 * B + b) * W + j] = s3o_f2h(act_bwd(activation, s3o_h2f(s3o_f2h(acc)), f));
        }
        /* through hidden matmuls, last to first */
        for (uint32_t k = 0; k + 1 < n_layers; k++) {
            const float* wk = wf + n_first + n_hid * (n_layers - 2 - k); /* [W_out, W_in] */
            for (uint32_t j = 0; j < W; j++) {
                float acc = 0;
                for (uint32_t o = 0; o < W; o++)
                    acc = fmaf(s3o_h2f(backward_buffer[((size_t)k * B + b) * W + o]), wk[(size_t)o * W + j], acc);
                const float f = s3o_h2f(forward_buffer[((size_t)(n_layers - 2 - k) * B + b) * W + j]);
                backward_buffer[((size_t)(k + 1) * B + b) * W + j] = s3o_f2h(act_bwd(activation, s3o_h2f(s3o_f2h(acc)), f));
            }
        }
        if (grad_inputs) {
            for (uint32_t i = 0; i < in_dim; i++) {
                float acc = 0;
                for (uint32_t o = 0; o < W; o++)
                    acc = fmaf(s3o_h2f(backward_buffer[((size_t)(n_layers - 1) * B + b) * W + o]), wf[(size_t)o * in_dim + i], acc);
                grad_inputs[(size_t)b * in_dim + i] = s3o_f2h(acc);
            }
        }
    }
    /* weight gradients: dW = dY^T X, summed over the batch */
    memset(grad_weights, 0, nw * sizeof(float));
    {
        float* gw_last = grad_weights + n_first + n_hid * (n_layers - 1);
#pragma omp parallel for schedule(static)
        for (int64_t o = 0; o < (int64_t)out_dim; o++)
            for (uint32_t b = 0; b < B; b++) {
                const float g = s3o_h2f(grad[(size_t)b * out_dim + o]);
                if (g == 0.0f) continue;
                for (uint32_t j = 0; j < W; j++)
                    gw_last[(size_t)o * W + j] += g * s3o_h2f(forward_buffer[((size_t)(n_layers - 1) * B + b) * W + j]);
            }
    }
    for (uint32_t k = 0; k + 1 < n_layers; k++) {
        float* gw = grad_weights + n_first + n_hid * (n_layers - 2 - k);
#pragma omp parallel for schedule(static)
        for (int64_t o = 0; o < (int64_t)W; o++)
            for (uint32_t b = 0; b < B; b++) {
                const float g = s3o_h2f(backward_buffer[((size_t)k * B + b) * W + o]);
                if (g == 0.0f) continue;
                for (uint32_t j = 0; j < W; j++)
                    gw[(size_t)o * W + j] += g * s3o_h2f(forward_buffer[((size_t)(n_layers - 2 - k) * B + b) * W + j]);
            }
    }
#pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < (int64_t)W; o++)
        for (uint32_t b = 0; b < B; b++) {
            const float g = s3o_h2f(backward_buffer[((size_t)(n_layers - 1) * B + b) * W + o]);
            if (g == 0.0f) continue;
            for (uint32_t i = 0; i < in_dim; i++)
                grad_weights[(size_t)o * in_dim + i] += g * s3o_h2f(inputs[(size_t)b * in_dim + i]);
        }
    free(wf);
}
