/*
 * orc_ldpc.c -- ORACLE (test infrastructure, not product): LDPC encode / sum-product decode.
 *
 * Restates lib/dtl/ldpc_enc.cc, lib/dtl/ldpc_dec.cc and the GNU Radio 3.10 gr-fec classes
 * they drive (alist, GF2Mat::get_G, cldpc, awgn_bp).  The GR sources are not in the
 * reference tree; the GR parts are an unverified restatement of GR 3.10 semantics
 *  -- "parity unpinned" for the permutation and for every BP iteration.
 */
#include "oracle.h"
#include "orc_math.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct orc_ldpc {
    int N, M, K, rank;
    int E;              /* number of edges                                           */
    int* row_ptr;       /* M+1 : edges of check c are [row_ptr[c], row_ptr[c+1])      */
    int* edge_var;      /* E   : variable (0-based) of each edge, mlist order         */
    int* col_ptr;       /* N+1 : var v's edges (nlist order) are col_edge[col_ptr[v]..)*/
    int* col_edge;      /* E   : index into the check-major edge arrays               */
    int* permute;       /* N   : column permutation recorded by get_G                 */
    uint8_t* parity;    /* rank x K : G[i][rank + j]                                  */
};

/* GR3.10 alist::read : "N M / maxcol maxrow / col weights / row weights / nlist / mlist",
 * each list line holds at least weight entries, 1-based, extra zero padding ignored. */
static int read_alist(const char* path, int* N, int* M, int** ncw, int** nrw, int*** nlist,
                      int*** mlist)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int bn, bm;
    if (fscanf(f, "%d %d %d %d", N, M, &bn, &bm) != 4) { fclose(f); return -1; }
    *ncw = (int*)calloc(*N, sizeof(int));
    *nrw = (int*)calloc(*M, sizeof(int));
    for (int i = 0; i < *N; i++) if (fscanf(f, "%d", &(*ncw)[i]) != 1) { fclose(f); return -1; }
    for (int i = 0; i < *M; i++) if (fscanf(f, "%d", &(*nrw)[i]) != 1) { fclose(f); return -1; }
    /* eat the rest of the weights line, then one text line per column / row */
    char* line = NULL; size_t cap = 0;
    if (getline(&line, &cap, f) < 0) { fclose(f); return -1; }
    *nlist = (int**)calloc(*N, sizeof(int*));
    *mlist = (int**)calloc(*M, sizeof(int*));
    for (int c = 0; c < *N; c++) {
        (*nlist)[c] = (int*)calloc((*ncw)[c] ? (*ncw)[c] : 1, sizeof(int));
        if (getline(&line, &cap, f) < 0) { fclose(f); return -1; }
        char* p = line;
        for (int e = 0; e < (*ncw)[c]; e++) (*nlist)[c][e] = (int)strtol(p, &p, 10);
    }
    for (int r = 0; r < *M; r++) {
        (*mlist)[r] = (int*)calloc((*nrw)[r] ? (*nrw)[r] : 1, sizeof(int));
        if (getline(&line, &cap, f) < 0) { fclose(f); return -1; }
        char* p = line;
        for (int e = 0; e < (*nrw)[r]; e++) (*mlist)[r][e] = (int)strtol(p, &p, 10);
    }
    free(line);
    fclose(f);
    return 0;
}

/* GR3.10 GF2Mat::get_G(permute, rank): Gauss-Jordan over GF(2) with column swaps; a row
 * with no pivot is rotated to the bottom and the active limit shrinks. */
static void gf2_get_G(uint8_t* G, int M, int N, int* permute, int* rank_out)
{
    int rank = 0;
    for (int c = 0; c < N; c++) permute[c] = c;
    int diag = 0, limit = M;
    uint8_t* tmp = (uint8_t*)malloc(N);
    while (diag < limit) {
        int non_zero = 0;
        for (int col = diag; col < N; col++) {
            if (G[diag * N + col] == 1) {
                non_zero = 1;
                rank++;
                if (col != diag) {
                    for (int r = 0; r < M; r++) { /* swap_cols */
                        uint8_t t = G[r * N + diag]; G[r * N + diag] = G[r * N + col]; G[r * N + col] = t;
                    }
                    int t = permute[diag]; permute[diag] = permute[col]; permute[col] = t;
                }
                break;
            }
        }
        if (non_zero) {
            for (int row = 0; row < M; row++) {
                if (row == diag) continue;
                if (G[row * N + diag] == 1) /* add_rows(row, diag) */
                    for (int c = 0; c < N; c++) G[row * N + c] ^= G[diag * N + c];
            }
            diag++;
        } else {
            memcpy(tmp, &G[diag * N], N);
            for (int row = diag; row < limit - 1; row++) memcpy(&G[row * N], &G[(row + 1) * N], N);
            memcpy(&G[(limit - 1) * N], tmp, N);
            limit--;
        }
    }
    free(tmp);
    *rank_out = rank;
}

orc_ldpc* orc_ldpc_load(const char* alist_path)
{
    int N, M, *ncw = NULL, *nrw = NULL, **nlist = NULL, **mlist = NULL;
    if (read_alist(alist_path, &N, &M, &ncw, &nrw, &nlist, &mlist) != 0) return NULL;
    orc_ldpc* c = (orc_ldpc*)calloc(1, sizeof(orc_ldpc));
    c->N = N; c->M = M;
    c->row_ptr = (int*)calloc(M + 1, sizeof(int));
    for (int r = 0; r < M; r++) c->row_ptr[r + 1] = c->row_ptr[r] + nrw[r];
    c->E = c->row_ptr[M];
    c->edge_var = (int*)calloc(c->E, sizeof(int));
    for (int r = 0; r < M; r++)
        for (int e = 0; e < nrw[r]; e++) c->edge_var[c->row_ptr[r] + e] = mlist[r][e] - 1;
    c->col_ptr = (int*)calloc(N + 1, sizeof(int));
    for (int v = 0; v < N; v++) c->col_ptr[v + 1] = c->col_ptr[v] + ncw[v];
    c->col_edge = (int*)calloc(c->col_ptr[N] ? c->col_ptr[N] : 1, sizeof(int));
    for (int v = 0; v < N; v++)
        for (int i = 0; i < ncw[v]; i++) {
            int chk = nlist[v][i] - 1, found = -1;
            for (int e = c->row_ptr[chk]; e < c->row_ptr[chk + 1]; e++)
                if (c->edge_var[e] == v) { found = e; break; }
            c->col_edge[c->col_ptr[v] + i] = found;
        }
    /* H (GR3.10 alist::get_matrix builds it from mlist) -> cldpc::set_alist */
    uint8_t* G = (uint8_t*)calloc((size_t)M * N, 1);
    for (int r = 0; r < M; r++)
        for (int e = 0; e < nrw[r]; e++) G[r * N + mlist[r][e] - 1] = 1;
    c->permute = (int*)calloc(N, sizeof(int));
    gf2_get_G(G, M, N, c->permute, &c->rank);
    c->K = N - c->rank;
    c->parity = (uint8_t*)calloc((size_t)c->rank * c->K + 1, 1);
    for (int i = 0; i < c->rank; i++)
        for (int j = 0; j < c->K; j++) c->parity[i * c->K + j] = G[i * N + c->rank + j];
    free(G);
    for (int v = 0; v < N; v++) free(nlist[v]);
    for (int r = 0; r < M; r++) free(mlist[r]);
    free(nlist); free(mlist); free(ncw); free(nrw);
    return c;
}

void orc_ldpc_free(orc_ldpc* c)
{
    if (!c) return;
    free(c->row_ptr); free(c->edge_var); free(c->col_ptr); free(c->col_edge);
    free(c->permute); free(c->parity); free(c);
}

int orc_ldpc_n(const orc_ldpc* c) { return c->N; }
int orc_ldpc_k(const orc_ldpc* c) { return c->K; }
int orc_ldpc_m(const orc_ldpc* c) { return c->M; }
int orc_ldpc_nedges(const orc_ldpc* c) { return c->E; }
void orc_ldpc_permute(const orc_ldpc* c, int* out) { memcpy(out, c->permute, c->N * sizeof(int)); }
void orc_ldpc_parity_matrix(const orc_ldpc* c, uint8_t* out)
{
    memcpy(out, c->parity, (size_t)c->rank * c->K);
}

/* GR3.10 cldpc::encode: x = [parity(rank) | data(K)], y[permute[i]] = x[i]; then
 * lib/dtl/ldpc_enc.cc:65-67 undoes the permutation (out[i] = y[permute[i]] = x[i]). */
void orc_ldpc_encode(const orc_ldpc* c, const uint8_t* data, uint8_t* out)
{
    for (int i = 0; i < c->rank; i++) {
        unsigned acc = 0;
        const uint8_t* row = &c->parity[i * c->K];
        for (int j = 0; j < c->K; j++) acc ^= (row[j] & data[j] & 1u);
        out[i] = (uint8_t)acc;
    }
    for (int j = 0; j < c->K; j++) out[c->rank + j] = data[j] & 1u;
}

static int is_codeword(const orc_ldpc* c, const uint8_t* est)
{
    for (int r = 0; r < c->M; r++) {
        unsigned s = 0;
        for (int e = c->row_ptr[r]; e < c->row_ptr[r + 1]; e++) s ^= est[c->edge_var[e]];
        if (s) return 0;
    }
    return 1;
}

/* lib/dtl/ldpc_dec.cc:62-71 + GR3.10 awgn_bp::{decode, rx_lr_calc, spa_initialize,
 * update_chks, update_vars, decision, is_codeword}; likelihood-ratio domain, double. */
int orc_ldpc_decode(const orc_ldpc* c, const float* llr, int max_it, float sigma,
                    uint8_t* out, int* nit)
{
    const int N = c->N, M = c->M, E = c->E;
    float* y = (float*)malloc(N * sizeof(float));
    uint8_t* est = (uint8_t*)malloc(N);
    /* ldpc_dec.cc:64-66 */
    for (int i = 0; i < N; i++) y[c->permute[i]] = -1 * llr[i];
    *nit = 0;
    /* awgn_bp::compute_init_estimate */
    for (int i = 0; i < N; i++) est[i] = (y[i] < 0) ? 1 : 0;
    if (!is_codeword(c, est)) {
        double* rx_lr = (double*)malloc(N * sizeof(double));
        double* lr = (double*)malloc(N * sizeof(double));
        double* Q = (double*)malloc(E * sizeof(double));
        double* R = (double*)malloc(E * sizeof(double));
        /* rx_lr_calc: float expression, exp evaluated on the float-rounded argument */
        for (int i = 0; i < N; i++) {
            float arg = (-1 * 2 * y[i]) / (sigma * sigma);
            rx_lr[i] = orc_exp((double)arg);
        }
        /* spa_initialize */
        for (int e = 0; e < E; e++) Q[e] = rx_lr[c->edge_var[e]];
        while (*nit < max_it) {
            *nit += 1;
            /* update_chks */
            for (int chk = 0; chk < M; chk++) {
                double product = 1.0;
                for (int e = c->row_ptr[chk]; e < c->row_ptr[chk + 1]; e++)
                    product = product * (2 / (1 + Q[e]) - 1);
                for (int e = c->row_ptr[chk]; e < c->row_ptr[chk + 1]; e++) {
                    double prd = product / (2 / (1 + Q[e]) - 1);
                    R[e] = (1 - prd) / (1 + prd);
                }
            }
            /* update_vars */
            for (int v = 0; v < N; v++) {
                double sum = rx_lr[v];
                for (int i = c->col_ptr[v]; i < c->col_ptr[v + 1]; i++) sum = sum * R[c->col_edge[i]];
                lr[v] = sum;
                for (int i = c->col_ptr[v]; i < c->col_ptr[v + 1]; i++)
                    Q[c->col_edge[i]] = sum / R[c->col_edge[i]];
            }
            /* decision */
            for (int v = 0; v < N; v++) est[v] = (lr[v] > 1) ? 1 : 0;
            if (is_codeword(c, est)) break;
        }
        free(rx_lr); free(lr); free(Q); free(R);
    }
    /* GR3.10 cldpc::get_systematic_bits: data[i] = in[permute[i + rank]] */
    for (int i = 0; i < c->K; i++) out[i] = est[c->permute[i + c->rank]];
    free(y); free(est);
    return c->K;
}

/* ---------------------------------------------------------------------------------------------
 * PRODUCTION ("fast") arithmetic of the decoder (SURVEY 7 "Hard parts" 2; NOT the reference's bits).
 * The same sum-product recursion in the same likelihood-ratio domain and in double, algebraically
 * identical to awgn_bp, with 2 divisions per CHECK per iteration instead of 4 per EDGE:
 *   - the d quotients of a row that share the form x_j / a_j take ONE division: prefix products
 *     P_j = a_0 .. a_j, I = 1 / P_{d-1}, then 1/a_j = I P_{j-1} walking back with I <- I a_j;
 *   - extrinsic products are prefix x suffix products instead of (full product) / own factor.
 * Everything is a plain IEEE double operation in the order written here (the HIP kernel executes the
 * same sequence; a row / column padded with the neutral element multiplies by exactly 1.0).  Bounded
 * messages keep every intermediate finite (the reference's form turns a saturated product into 0/inf and then
 * NaN, whose decision is 0 whatever was sent -- its error floor): channel LLR clamped to +-40, check-node
 * products to +-(1 - 2^-20) (check-to-variable ratios in [2^-21, 2^21], LLR within +-14.6: the clipping also
 * helps in the waterfall).  Variable-to-check ratios then stay below 2^121 for column weights <= 4 and a row's
 * product of (1 + Q) below 2^968 for row weights <= 8 (the product's kernel limits).  A NaN soft bit is taken as
 * -max (what fmin(fmax()) of the C library returns; written as comparisons so that no compiler substitutes a clamp
 * instruction with another NaN rule); fmin/fmax on the products return the non-NaN operand.
 * ------------------------------------------------------------------------------------------- */
#define ORC_FAST_LLR_MAX 40.0f
#define ORC_FAST_P_MAX (1.0 - 0x1p-20)
#define ORC_FAST_MAX_DEG 64

/* out[j] = num[j] / den[j], j < d, with one division (see above) */
static void batch_quotients(const double* num, const double* den, int d, double* out)
{
    double P[ORC_FAST_MAX_DEG];
    P[0] = den[0];
    for (int j = 1; j < d; j++) P[j] = P[j - 1] * den[j];
    double I = 1 / P[d - 1];
    for (int j = d - 1; j >= 1; j--) {
        out[j] = num[j] * (I * P[j - 1]);
        I = I * den[j];
    }
    out[0] = num[0] * I;
}

int orc_ldpc_decode_fast(const orc_ldpc* c, const float* llr, int max_it, float sigma,
                         uint8_t* out, int* nit)
{
    const int N = c->N, M = c->M, E = c->E;
    float* y = (float*)malloc(N * sizeof(float));
    uint8_t* est = (uint8_t*)malloc(N);
    for (int i = 0; i < N; i++) y[c->permute[i]] = -1 * llr[i];
    *nit = 0;
    for (int i = 0; i < N; i++) est[i] = (y[i] < 0) ? 1 : 0;
    if (!is_codeword(c, est)) {
        double* rx_lr = (double*)malloc(N * sizeof(double));
        double* Q = (double*)malloc(E * sizeof(double));
        double* R = (double*)malloc(E * sizeof(double));
        for (int i = 0; i < N; i++) {
            float arg = (-1 * 2 * y[i]) / (sigma * sigma);
            arg = arg > ORC_FAST_LLR_MAX ? ORC_FAST_LLR_MAX : (arg >= -ORC_FAST_LLR_MAX ? arg : -ORC_FAST_LLR_MAX); /* NaN -> -max */
            rx_lr[i] = orc_exp((double)arg);
        }
        for (int e = 0; e < E; e++) Q[e] = rx_lr[c->edge_var[e]];
        while (*nit < max_it) {
            *nit += 1;
            for (int chk = 0; chk < M; chk++) {
                const int e0 = c->row_ptr[chk], d = c->row_ptr[chk + 1] - e0;
                double a[ORC_FAST_MAX_DEG], b[ORC_FAST_MAX_DEG], t[ORC_FAST_MAX_DEG];
                double pre[ORC_FAST_MAX_DEG], suf[ORC_FAST_MAX_DEG];
                if (d <= 0 || d > ORC_FAST_MAX_DEG) continue;
                for (int j = 0; j < d; j++) { a[j] = 1 + Q[e0 + j]; b[j] = 1 - Q[e0 + j]; }
                batch_quotients(b, a, d, t);              /* t_j = (1 - Q_j) / (1 + Q_j) = 2/(1+Q_j) - 1 */
                pre[0] = 1.0;
                for (int j = 1; j < d; j++) pre[j] = pre[j - 1] * t[j - 1];
                suf[d - 1] = 1.0;
                for (int j = d - 2; j >= 0; j--) suf[j] = suf[j + 1] * t[j + 1];
                for (int j = 0; j < d; j++) {
                    double p = pre[j] * suf[j];           /* product of the other edges' t */
                    p = fmin(fmax(p, -ORC_FAST_P_MAX), ORC_FAST_P_MAX);
                    a[j] = 1 + p; b[j] = 1 - p;
                }
                batch_quotients(b, a, d, &R[e0]);         /* R_j = (1 - p_j) / (1 + p_j) */
            }
            for (int v = 0; v < N; v++) {
                const int i0 = c->col_ptr[v], dv = c->col_ptr[v + 1] - i0;
                double pre[ORC_FAST_MAX_DEG + 1], suf[ORC_FAST_MAX_DEG];
                if (dv > ORC_FAST_MAX_DEG) continue;
                pre[0] = rx_lr[v];
                for (int i = 0; i < dv; i++) pre[i + 1] = pre[i] * R[c->col_edge[i0 + i]];
                if (dv > 0) suf[dv - 1] = 1.0;
                for (int i = dv - 2; i >= 0; i--) suf[i] = suf[i + 1] * R[c->col_edge[i0 + i + 1]];
                for (int i = 0; i < dv; i++) Q[c->col_edge[i0 + i]] = pre[i] * suf[i];
                est[v] = (pre[dv] > 1) ? 1 : 0;           /* decision on the full product */
            }
            if (is_codeword(c, est)) break;
        }
        free(rx_lr); free(Q); free(R);
    }
    for (int i = 0; i < c->K; i++) out[i] = est[c->permute[i + c->rank]];
    free(y); free(est);
    return c->K;
}

int orc_ldpc_decode_arith(const orc_ldpc* c, const float* llr, int max_it, float sigma, int arith,
                          uint8_t* out, int* nit)
{
    return arith ? orc_ldpc_decode_fast(c, llr, max_it, sigma, out, nit)
                 : orc_ldpc_decode(c, llr, max_it, sigma, out, nit);
}

struct batch_arg {
    int arith;

    const orc_ldpc* c; const float* llr; int lo, hi, max_it; float sigma; uint8_t* out; int* nit;
};
static void* batch_worker(void* p)
{
    struct batch_arg* a = (struct batch_arg*)p;
    for (int i = a->lo; i < a->hi; i++)
        orc_ldpc_decode_arith(a->c, a->llr + (size_t)i * a->c->N, a->max_it, a->sigma, a->arith,
                              a->out + (size_t)i * a->c->K, &a->nit[i]);
    return NULL;
}

void orc_ldpc_decode_batch(const orc_ldpc* c, const float* llr, int ncw, int max_it,
                           float sigma, uint8_t* out_bits, int* nit, int n_threads)
{
    orc_ldpc_decode_batch_arith(c, llr, ncw, max_it, sigma, 0, out_bits, nit, n_threads);
}

void orc_ldpc_decode_batch_arith(const orc_ldpc* c, const float* llr, int ncw, int max_it,
                                 float sigma, int arith, uint8_t* out_bits, int* nit, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    pthread_t th[64];
    struct batch_arg args[64];
    for (int t = 0; t < n_threads; t++) {
        args[t] = (struct batch_arg){ arith, c, llr, (int)((long)ncw * t / n_threads),
                                      (int)((long)ncw * (t + 1) / n_threads), max_it, sigma,
                                      out_bits, nit };
        if (n_threads == 1) batch_worker(&args[t]);
        else pthread_create(&th[t], NULL, batch_worker, &args[t]);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}
