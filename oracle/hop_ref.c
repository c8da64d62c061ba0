/*
 * oracle/hop_ref.c -- scalar C restatement of the message-passing hop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the HIP
 * path and the single-core "port" CPU baseline of bench.py; never linked into
 * or called from the product (deformcontact_amd/).
 *
 * Parity unpinned at the PyG boundary: the arithmetic lives in torch_geometric
 * 2.5.2 (/root/reference/environment.yml:76), absent from /root/reference and
 * from this image.  Restated from PyG's published algorithm; reference call
 * sites: models/model.py:71,77 (conv(x, edge_index)), train.py:36-38
 * (Batch.from_data_list defines the edge order).
 *
 *   ref_csr_build  - stable counting sort of edges by edge_index[key_row]
 *                    (== np.argsort(kind="stable")): the order scatter_add_
 *                    visits the contributions of one destination.
 *   ref_gcn_norm   - gcn_conv.py: gcn_norm(add_self_loops=False): in-degree by
 *                    scatter of ones over edge_index[1], deg^-1/2 (inf -> 0),
 *                    w_e = dis[row_e] * 1 * dis[col_e].
 *   ref_hop        - message_passing.py propagate(aggr="add") with
 *                    message = w_e * x_j: walk edges in input order,
 *                    y[col_e,:] += w_e * x[row_e,:]  (fp32, one rounding for
 *                    the product and one for the add, as index_select -> mul ->
 *                    scatter_add_ does).
 *   ref_tag_linear - tag_conv.py: out = sum_k x_k W_k^T + b as k-ordered
 *                    F.linear calls added left to right.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int ref_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int key_row,
                  int32_t *ptr, int32_t *other, int32_t *perm)
{
    const int64_t *key = edge_index + (key_row ? E : 0);
    const int64_t *oth = edge_index + (key_row ? 0 : E);
    memset(ptr, 0, sizeof(int32_t) * (size_t)(N + 1));
    for (int64_t e = 0; e < E; ++e) {
        if (key[e] < 0 || key[e] >= N || oth[e] < 0 || oth[e] >= N) return -1;
        ptr[key[e] + 1]++;
    }
    for (int64_t i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    if (!cur) return -2;
    memcpy(cur, ptr, sizeof(int32_t) * (size_t)N);
    for (int64_t e = 0; e < E; ++e) {
        int32_t p = cur[key[e]]++;
        other[p] = (int32_t)oth[e];
        perm[p] = (int32_t)e;
    }
    free(cur);
    return 0;
}

void ref_gcn_norm(const int64_t *edge_index, int64_t E, int64_t N, float *w)
{
    const int64_t *row = edge_index, *col = edge_index + E;
    float *deg = (float *)calloc((size_t)(N > 0 ? N : 1), sizeof(float));
    for (int64_t e = 0; e < E; ++e) deg[col[e]] += 1.0f;
    for (int64_t i = 0; i < N; ++i) deg[i] = deg[i] > 0.0f ? 1.0f / sqrtf(deg[i]) : 0.0f;
    for (int64_t e = 0; e < E; ++e) w[e] = deg[row[e]] * 1.0f * deg[col[e]];
    free(deg);
}

void ref_hop(const int64_t *edge_index, const float *w, const float *x, int64_t ldx,
             int64_t E, int64_t N, int64_t F, float *y, int64_t ldy)
{
    const int64_t *row = edge_index, *col = edge_index + E;
    for (int64_t i = 0; i < N; ++i) memset(y + i * ldy, 0, sizeof(float) * (size_t)F);
    for (int64_t e = 0; e < E; ++e) {
        const float *xs = x + row[e] * ldx;
        float *yd = y + col[e] * ldy;
        const float we = w[e];
        for (int64_t f = 0; f < F; ++f) {
            volatile float m = we * xs[f]; /* keep mul and add separately rounded */
            yd[f] += m;
        }
    }
}

/* bf16-stored features, fp32 accumulation: every term is
 * fp32(w_e) * fp32(x[row_e,f]) added to an fp32 running sum in edge order; y_is_f32 selects an
 * fp32 result or ONE round-to-nearest-even to bf16 at the end. */
static float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
void ref_hop_bf16(const int64_t *edge_index, const float *w, const uint16_t *x, int64_t ldx,
                  int64_t E, int64_t N, int64_t F, void *y, int64_t ldy, int y_is_f32)
{
    const int64_t *row = edge_index, *col = edge_index + E;
    float *acc = (float *)calloc((size_t)(N * F > 0 ? N * F : 1), sizeof(float));
    for (int64_t e = 0; e < E; ++e) {
        const uint16_t *xs = x + row[e] * ldx;
        float *yd = acc + col[e] * F;
        const float we = w ? w[e] : 1.0f;
        for (int64_t f = 0; f < F; ++f) {
            volatile float m = we * bf16_to_f32(xs[f]);
            yd[f] += m;
        }
    }
    for (int64_t i = 0; i < N; ++i)
        for (int64_t f = 0; f < F; ++f) {
            if (y_is_f32) ((float *)y)[i * ldy + f] = acc[i * F + f];
            else ((uint16_t *)y)[i * ldy + f] = f32_to_bf16_rne(acc[i * F + f]);
        }
    free(acc);
}

/* out[N,Fo] = sum_k xs[k][N,Fi] . W[k][Fo,Fi]^T + b ; K1 = K+1 operand pairs */
void ref_tag_linear(const float *const *xs, const float *const *Ws, const float *b,
                    int K1, int64_t N, int64_t Fi, int64_t Fo, float *out)
{
    for (int64_t i = 0; i < N; ++i)
        for (int64_t o = 0; o < Fo; ++o) {
            float acc = 0.0f;
            for (int k = 0; k < K1; ++k) {
                const float *xr = xs[k] + i * Fi, *wr = Ws[k] + o * Fi;
                float s = 0.0f;
                for (int64_t f = 0; f < Fi; ++f) s += xr[f] * wr[f];
                acc = (k == 0) ? s : acc + s;
            }
            out[i * Fo + o] = b ? acc + b[o] : acc;
        }
}
