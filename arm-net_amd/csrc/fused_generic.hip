// fused_generic.hip — shape-agnostic fused ARM block (any F, E, O).  gfx950.
//
// Correctness-first companion of fused_mfma.hip: covers the shapes that kernel does not
// specialise (e.g. Frappe nfield=10/nemb=10/nhid=10) with the same math and the same solvers.
// One 128-thread block stages S samples' value-scaled rows in LDS, then every thread owns one
// (sample, neuron) row: gates -> LDS column -> sparse map in place -> weights -> interaction.
#include "armnet_common.h"

namespace armnet {

constexpr int GEN_TPB = 128;

template <typename IdT, bool FROM_ROWS>
__global__ void __launch_bounds__(GEN_TPB)
fused_generic_kernel(FusedArgs a, int S) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int F = a.F, E = a.E, O = a.O;
    float* xs = lds;                                   // [S][F][E]   value-scaled rows
    float* gs = lds + (((size_t)S * F * E + 3) & ~(size_t)3);   // [F][GEN_TPB+1] one column per thread
    constexpr int GS = GEN_TPB + 1;
    const IdT* ids = reinterpret_cast<const IdT*>(a.ids);
    const int tid = threadIdx.x;

    for (int64_t b0 = (int64_t)blockIdx.x * S; b0 < a.B; b0 += (int64_t)gridDim.x * S) {
        const int ns = (int)((a.B - b0) < S ? (a.B - b0) : S);
        __syncthreads();
        // a2 + a3: clamp, gather, scale  (armnet_1h.py:81, layers.py:20-21)
        for (int k = tid; k < ns * F * E; k += GEN_TPB) {
            const int sf = k / E, e = k - sf * E;
            const int64_t gi = b0 * F + sf;
            const float vraw = a.vals[gi];
            const float v = clamp_val(vraw);
            float row;
            if constexpr (FROM_ROWS) {
                row = a.rows[gi * E + e];
            } else {
                bool bad;
                const uint32_t id = load_id_checked(ids + gi, a.nfeat, bad);
                if (bad && a.id_status && e == 0) flag_bad_id(a.id_status);
                row = a.table[(size_t)id * E + e];
            }
            xs[k] = row * v;
            if (e == 0 && (a.flags & ARMNET_F_WRITE_CLAMPED_VALS) && v != vraw) a.vals[gi] = v;
        }
        __syncthreads();
        if (a.model == MODEL_AFN) {
            // afn.py:63-66: z[o,e] = sum_f W[o,f] * (log(x[f,e]) * s_f + t_f) + b[o]; out = afn_bn(exp(z))
            for (int k = tid; k < ns * F * E; k += GEN_TPB) {
                const int f = (k / E) % F;
                xs[k] = fmaf(logf(xs[k]), a.emb_scale[f], a.emb_shift[f]);
            }
            __syncthreads();
            for (int r = tid; r < ns * O; r += GEN_TPB) {
                const int s = r / O, o = r - s * O;
                const float* x = xs + (size_t)s * F * E;
                const float sc = a.bn_scale[o], sh = a.bn_shift[o], bias = a.lin_bias[o];
                float* dst = a.out + ((b0 + s) * O + o) * (int64_t)E;
                for (int e = 0; e < E; ++e) {
                    float acc = 0.f;
                    for (int f = 0; f < F; ++f) acc = fmaf(x[f * E + e], a.values[(size_t)o * F + f], acc);
                    dst[e] = fmaf(exp_accurate(acc + bias), sc, sh);
                }
            }
            continue;
        }
        for (int r = tid; r < ns * O; r += GEN_TPB) {
            const int s = r / O, o = r - s * O;
            const float* x = xs + (size_t)s * F * E;
            const float* qf = a.q_fold + (size_t)o * E;
            float* col = gs + tid;
            // a4+a5 folded: g[f] = sum_e x[f,e] * q_fold[o,e]
            for (int f = 0; f < F; ++f) {
                float acc = 0.f;
                for (int e = 0; e < E; ++e) acc = fmaf(x[f * E + e], qf[e], acc);
                col[f * GS] = acc;
            }
            if (a.model == MODEL_GC_ARM) {
                // gc_arm.py:37-41: global context = the gates of the field sum = the sum of the gates (real arithmetic)
                float gc = 0.f;
                for (int f = 0; f < F; ++f) gc += col[f * GS];
                for (int f = 0; f < F; ++f) col[f * GS] += gc;
            }
            sparse_map_row(col, GS, F, a.cfg);                            // a6
            for (int f = 0; f < F; ++f) col[f * GS] *= a.values[(size_t)o * F + f];   // a7
            const float sc = a.bn_scale[o], sh = a.bn_shift[o];
            float* dst = a.out + ((b0 + s) * O + o) * (int64_t)E;
            if (a.model == MODEL_GC_ARM) {
                // gc_arm.py:89-94: arm = sum_f w[f] * emb_bn(exp(x))[f,e]   (no outer exp), then arm_bn
                for (int e = 0; e < E; ++e) {
                    float acc = 0.f;
                    for (int f = 0; f < F; ++f)
                        acc = fmaf(col[f * GS], fmaf(exp_accurate(x[f * E + e]), a.emb_scale[f], a.emb_shift[f]), acc);
                    dst[e] = fmaf(acc, sc, sh);
                }
                continue;
            }
            for (int e = 0; e < E; ++e) {                                 // a8 + a9
                float acc = 0.f;
                for (int f = 0; f < F; ++f) acc = fmaf(col[f * GS], x[f * E + e], acc);
                dst[e] = fmaf(exp_accurate(acc), sc, sh);
            }
        }
    }
}

int launch_fused_generic(const FusedArgs& a, hipStream_t s) {
    if (a.B == 0) return ARMNET_OK;
    // samples per block: enough rows to occupy the block's threads, bounded by LDS
    int S = (GEN_TPB + a.O - 1) / a.O;
    if (S < 1) S = 1;
    const size_t gates_bytes = (size_t)a.F * (GEN_TPB + 1) * sizeof(float);
    auto need = [&](int s_) { return ((((size_t)s_ * a.F * a.E + 3) & ~(size_t)3)) * sizeof(float) + gates_bytes; };
    while (S > 1 && need(S) > 64 * 1024) --S;
    if (need(S) > 64 * 1024) return ARMNET_ERR_UNSUPPORTED;
    int64_t grid = (a.B + S - 1) / S;
    if (grid > 256 * 8) grid = 256 * 8;
    const size_t lds = need(S);
    const bool from_rows = a.rows != nullptr;
    if (from_rows)
        fused_generic_kernel<int64_t, true><<<(int)grid, GEN_TPB, lds, s>>>(a, S);
    else if (a.id_type == ARMNET_ID_I64)
        fused_generic_kernel<int64_t, false><<<(int)grid, GEN_TPB, lds, s>>>(a, S);
    else
        fused_generic_kernel<int32_t, false><<<(int)grid, GEN_TPB, lds, s>>>(a, S);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

}  // namespace armnet
