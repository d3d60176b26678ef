/*
 * mpa_oracle.c -- CPU restatement of the reference's index-producing point-set ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may call, link or load this
 * file: it is the checker for the HIP kernels (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  It restates, in plain scalar C with every rounding
 * step written out, the arithmetic the reference performs through PyTorch-CPU ATen for
 *
 *   square_distance        modules/pointnet2_utils.py:190-209  (== repsurface_utils.py:129-148)
 *   knn_point              modules/pointnet2_utils.py:211-222  (== repsurface_utils.py:193-204)
 *   farthest_point_sample  modules/pointnet2_utils.py:84-109   (== repsurface_utils.py:150-172)
 *   query_ball_point       modules/pointnet2_utils.py:112-134
 *   3-NN of PointNetFeaturePropagation   modules/pointnet2_utils.py:899-901
 *
 * The rounding models are those of SURVEY.md Appendix A (A1..A6).  Parity is PINNED:
 * tests/test_oracle_golden.py checks every function here bit-for-bit against golden
 * vectors produced by importing the reference itself (tests/golden/make_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).  Contraction must stay
 * off: products and sums below are separate roundings unless fmaf() is written.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* A2: torch.sum(x**2, -1) on CPU.  Squares are rounded on their own (no FMA).
 * C < 8: plain left-to-right sum.  C a multiple of 8: the C squares form C/8 vectors of
 * 8 lanes; min(4, C/8) lane-wise accumulators, accumulator a taking vectors a, a+A, ...;
 * accumulators are combined ((a0+a1)+a2)+a3 lane-wise, then lanes 0..7 left to right.
 * Other C: vector part as above over floor(C/8)*8, then the tail is added left to right
 * (validated against the reference's farthest_point_sample at C = 10: tests/golden/round2.npz). */
static float sum_sq(const float *x, int C)
{
    if (C < 8) {
        float s = x[0] * x[0];
        for (int c = 1; c < C; ++c) {
            float sq = x[c] * x[c];
            s = s + sq;
        }
        return s;
    }
    int nv = C / 8;
    int A = nv < 4 ? nv : 4;
    float acc[4][8];
    for (int a = 0; a < A; ++a)
        for (int l = 0; l < 8; ++l) {
            float v = x[a * 8 + l];
            acc[a][l] = v * v;
        }
    for (int v = A; v < nv; ++v) {
        int a = v % A;
        for (int l = 0; l < 8; ++l) {
            float e = x[v * 8 + l];
            float sq = e * e;
            acc[a][l] = acc[a][l] + sq;
        }
    }
    float lane[8];
    for (int l = 0; l < 8; ++l) {
        float s = acc[0][l];
        for (int a = 1; a < A; ++a)
            s = s + acc[a][l];
        lane[l] = s;
    }
    float s = lane[0];
    for (int l = 1; l < 8; ++l)
        s = s + lane[l];
    for (int c = nv * 8; c < C; ++c) {
        float sq = x[c] * x[c];
        s = s + sq;
    }
    return s;
}

/* A1: the bmm dot over channels is a sequential FMA chain in channel order. */
static float dot_chain(const float *q, const float *b, int C)
{
    float acc = q[0] * b[0];
    for (int c = 1; c < C; ++c)
        acc = fmaf(q[c], b[c], acc);
    return acc;
}

/* A3: d = fl(-2*dot); d = fl(d + |q|^2); d = fl(d + |b|^2). */
static inline float sqdist_from(float dot, float qn, float bn)
{
    float d = -2.0f * dot;
    d = d + qn;
    d = d + bn;
    return d;
}

/* square_distance(src=query [S,C], dst=base [N,C]) -> out [S,N], for B batches. */
void orc_square_distance(const float *query, const float *base, int B, int S, int N, int C,
                         float *out)
{
    float *bn = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *Q = query + (size_t)b * S * C;
        const float *P = base + (size_t)b * N * C;
        float *O = out + (size_t)b * S * N;
        for (int n = 0; n < N; ++n)
            bn[n] = sum_sq(P + (size_t)n * C, C);
        for (int s = 0; s < S; ++s) {
            float qn = sum_sq(Q + (size_t)s * C, C);
            for (int n = 0; n < N; ++n)
                O[(size_t)s * N + n] = sqdist_from(dot_chain(Q + (size_t)s * C, P + (size_t)n * C, C), qn, bn[n]);
        }
    }
    free(bn);
}

/* A4: knn_point(K, xyz=base, new_xyz=query) -> (dist [S,K], idx [S,K]) ascending by
 * (distance, index).  The reference's topk is not stable on exact ties; this project
 * defines lowest-index-first, which is what a stable ascending sort gives and what the
 * reference produced on all tie-free golden inputs. */
void orc_knn(const float *base, const float *query, int B, int N, int S, int C, int K,
             float *out_dist, int64_t *out_idx)
{
    float *bn = (float *)malloc(sizeof(float) * (size_t)N);
    float *bd = (float *)malloc(sizeof(float) * (size_t)K);
    int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)K);
    for (int b = 0; b < B; ++b) {
        const float *Q = query + (size_t)b * S * C;
        const float *P = base + (size_t)b * N * C;
        for (int n = 0; n < N; ++n)
            bn[n] = sum_sq(P + (size_t)n * C, C);
        for (int s = 0; s < S; ++s) {
            float qn = sum_sq(Q + (size_t)s * C, C);
            int cnt = 0;
            for (int n = 0; n < N; ++n) {
                float d = sqdist_from(dot_chain(Q + (size_t)s * C, P + (size_t)n * C, C), qn, bn[n]);
                if (cnt < K) {
                    int p = cnt++;
                    while (p > 0 && d < bd[p - 1]) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                    bd[p] = d; bi[p] = n;
                } else if (d < bd[K - 1]) {
                    int p = K - 1;
                    while (p > 0 && d < bd[p - 1]) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                    bd[p] = d; bi[p] = n;
                }
            }
            for (int k = 0; k < K; ++k) {
                out_dist[((size_t)b * S + s) * K + k] = k < cnt ? bd[k] : INFINITY;
                out_idx[((size_t)b * S + s) * K + k] = k < cnt ? bi[k] : -1;
            }
        }
    }
    free(bn); free(bd); free(bi);
}

/* A5: farthest_point_sample(xyz [B,N,3], S) with caller-supplied start indices (the
 * reference draws them with torch.randint on the CPU generator, pointnet2_utils.py:96). */
void orc_fps(const float *xyz, int B, int N, int C, int S, const int64_t *start, int64_t *out_idx)
{
    float *dist = (float *)malloc(sizeof(float) * (size_t)N);
    float *diff = (float *)malloc(sizeof(float) * (size_t)C);
    for (int b = 0; b < B; ++b) {
        const float *P = xyz + (size_t)b * N * C;
        for (int n = 0; n < N; ++n)
            dist[n] = 1e10f;
        int64_t far = start[b];
        for (int i = 0; i < S; ++i) {
            out_idx[(size_t)b * S + i] = far;
            const float *c = P + (size_t)far * C;
            float best = -INFINITY;
            int64_t besti = 0;
            for (int n = 0; n < N; ++n) {
                const float *p = P + (size_t)n * C;
                /* torch.sum((xyz - centroid)**2, -1): differences rounded, then the A2 sum of
                 * squares (C == 3: (dx0^2 + dx1^2) + dx2^2; C >= 8: the 8-lane x 4-accumulator order) */
                for (int k = 0; k < C; ++k)
                    diff[k] = p[k] - c[k];
                float dd = sum_sq(diff, C);
                if (dd < dist[n])
                    dist[n] = dd;
                if (dist[n] > best) { best = dist[n]; besti = n; }   /* first maximum */
            }
            far = besti;
        }
    }
    free(dist);
    free(diff);
}

/* A6: query_ball_point(radius, nsample, xyz=base, new_xyz=query): first nsample base
 * indices (ascending) with not (d > fl32(r^2)); padded with the first hit; a row with no
 * hit is all N (as the reference leaves it). */
void orc_ball_query(const float *base, const float *query, int B, int N, int S, int C,
                    float radius2, int nsample, int64_t *out_idx)
{
    float *bn = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *Q = query + (size_t)b * S * C;
        const float *P = base + (size_t)b * N * C;
        for (int n = 0; n < N; ++n)
            bn[n] = sum_sq(P + (size_t)n * C, C);
        for (int s = 0; s < S; ++s) {
            float qn = sum_sq(Q + (size_t)s * C, C);
            int64_t *o = out_idx + ((size_t)b * S + s) * nsample;
            int cnt = 0;
            for (int n = 0; n < N && cnt < nsample; ++n) {
                float d = sqdist_from(dot_chain(Q + (size_t)s * C, P + (size_t)n * C, C), qn, bn[n]);
                if (!(d > radius2))
                    o[cnt++] = n;
            }
            int64_t first = cnt > 0 ? o[0] : N;
            for (int k = cnt; k < nsample; ++k)
                o[k] = first;
        }
    }
    free(bn);
}

/* 3-NN of PointNetFeaturePropagation: square_distance(xyz1=query, xyz2=base).sort()[:3]. */
void orc_three_nn(const float *query, const float *base, int B, int Nq, int Nb, int C,
                  float *out_dist, int64_t *out_idx)
{
    orc_knn(base, query, B, Nb, Nq, C, 3, out_dist, out_idx);
}

int orc_version(void) { return 1; }
