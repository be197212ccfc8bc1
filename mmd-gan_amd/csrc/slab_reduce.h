// The slab reduction of the Winograd-domain weight gradients, as a piece of device code that two kinds of launch run:
//   * slab_reduce_kernel (conv_wino2.hip): the stand-alone pass - one bandwidth-only launch behind its producer;
//   * the PROLOGUE of the next weight-gradient launch on the same stream (mmdgan_wgrad_defer): every workgroup of that
//     launch first sums its 1/grid share of the previous layer's slabs, then does its own work.  In the training step the
//     stand-alone pass sat in the weight-gradient queue between two MFMA kernels and was starved of CUs by the main queue's
//     persistent kernels (10 us alone, 22-58 us in the step: 243 us of that queue per CIFAR step, profiles/r04_step_timeline.txt);
//     as a prologue it is ~130 KB of reads per workgroup and needs no fence: the kernel boundary publishes the slabs.
// Both run slab_reduce_range below, so a gradient is bit-identical whichever launch summed it: the order of the additions is a
// function of nsplit alone - nsplit >= 8: four interleaved partial sums (slabs s = q mod 4, ascending) combined as
// (p0 + p1) + (p2 + p3); fewer slabs: one ascending sum.
#pragma once
#include "common.h"

namespace mmdgan {

struct SlabReduceArgs {           // passed to kernels by value; nsplit == 0: nothing to do
    const float4 *part;           // [nsplit][n4] partial weight gradients
    int nsplit;
    long n4;
    float4 *dw;                   // [n4]
    const float4 *dbpart;         // [nsplit][k4] partial bias gradients (k4 == 0: none)
    long k4;
    float4 *dbias;
    const float4 *wdot;           // optional: dot[0] += <dw, wdot> (the spectral-norm fix-up's scalar)
    float *dot;
};

constexpr int kSlabLane = 14;     // slabs of an element per lane and pass in the quad form (x 4 lanes = 56: the largest split of a one-round grid)
constexpr int kSlabQuadMin = 8;   // from this many slabs on, four lanes share an element (each a quarter of the slabs)

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// elements [e0, e1) of the concatenated index space [0, n4) (weights) ++ [n4, n4 + k4) (bias) by `nthreads` threads
// (a multiple of 64; in quad mode a multiple of 4 lanes handles nthreads / 4 elements per trip).  Returns this thread's part
// of <dw, wdot> (0 without wdot); the caller reduces it over the workgroup and adds it to *dot.
template <int U>                 // elements of a thread in flight in the few-slabs form (x up to 7 loads each)
__device__ __forceinline__ double slab_reduce_range(const SlabReduceArgs &a, long e0, long e1, int tid, int nthreads) {
    double acc = 0;
    const long total = a.n4 + a.k4;
    if (e1 > total) e1 = total;
    if (a.nsplit >= kSlabQuadMin) {
        // two elements of the lane quad and up to fourteen slabs of each in flight per trip: a workgroup's share of a 56-slab
        // reduction (~160 elements) is ONE round of loads (first cut: eight in flight, one element per trip - four dependent
        // rounds, 5-6 us in front of every weight-gradient launch of a chain)
        constexpr int H = U >= 4 ? 2 : 1;                                     // (the stand-alone pass keeps a small register footprint)
        const int q = tid & 3, step = nthreads >> 2;
        for (long base = e0; base < e1; base += H * step) {                  // (uniform trip count: the shuffles below need every lane)
            float4 p[H];
            bool live[H], isw[H];
            long el[H];
            for (int s0 = 0; s0 < a.nsplit; s0 += 4 * kSlabLane) {           // (chunks of 56 slabs: one pass for every shape of the step)
                float4 v[H][kSlabLane];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    long e = base + h * step + (tid >> 2);
                    live[h] = e < e1;
                    isw[h] = e < a.n4;
                    const float4 *src = isw[h] ? a.part : a.dbpart;
                    const long slab = isw[h] ? a.n4 : a.k4;
                    if (!isw[h]) e -= a.n4;
                    el[h] = e;
                    if (s0 == 0) p[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < kSlabLane; ++i) {
                        const int sl = s0 + q + 4 * i;
                        if (live[h] && sl < a.nsplit) v[h][i] = src[(long)sl * slab + e];
                    }
                }
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int i = 0; i < kSlabLane; ++i)
                        if (live[h] && s0 + q + 4 * i < a.nsplit) p[h] = f4add(p[h], v[h][i]);     // ascending s = q, q + 4, ...: the canonical order
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float4 o;                                                    // (p0 + p1) + (p2 + p3), the same bits in all four lanes
                o.x = p[h].x + __shfl_xor(p[h].x, 1, 64); o.y = p[h].y + __shfl_xor(p[h].y, 1, 64);
                o.z = p[h].z + __shfl_xor(p[h].z, 1, 64); o.w = p[h].w + __shfl_xor(p[h].w, 1, 64);
                o.x += __shfl_xor(o.x, 2, 64); o.y += __shfl_xor(o.y, 2, 64);
                o.z += __shfl_xor(o.z, 2, 64); o.w += __shfl_xor(o.w, 2, 64);
                if (live[h] && q == 0) {
                    (isw[h] ? a.dw : a.dbias)[el[h]] = o;
                    if (a.wdot && isw[h]) {
                        const float4 wv = a.wdot[el[h]];
                        acc += (double)o.x * wv.x + (double)o.y * wv.y + (double)o.z * wv.z + (double)o.w * wv.w;
                    }
                }
            }
        }
        return acc;
    }
    // few slabs: one thread per element, U elements of the thread in flight
    for (long base = e0 + tid; base < e1; base += (long)U * nthreads) {
        float4 v[U][kSlabQuadMin - 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long e = base + (long)u * nthreads;
            if (e >= e1) continue;
            const bool is_w = e < a.n4;
            const float4 *src = is_w ? a.part : a.dbpart;
            const long slab = is_w ? a.n4 : a.k4;
            if (!is_w) e -= a.n4;
#pragma unroll
            for (int s = 0; s < kSlabQuadMin - 1; ++s)
                if (s < a.nsplit) v[u][s] = src[(long)s * slab + e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long e = base + (long)u * nthreads;
            if (e >= e1) continue;
            const bool is_w = e < a.n4;
            if (!is_w) e -= a.n4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < kSlabQuadMin - 1; ++s)
                if (s < a.nsplit) o = f4add(o, v[u][s]);
            (is_w ? a.dw : a.dbias)[e] = o;
            if (a.wdot && is_w) {
                const float4 wv = a.wdot[e];
                acc += (double)o.x * wv.x + (double)o.y * wv.y + (double)o.z * wv.z + (double)o.w * wv.w;
            }
        }
    }
    return acc;
}

// the share of workgroup `wg` of `nwg` (all `nthreads` threads of the workgroup call it; `red`: >= nthreads / 64 doubles of
// LDS the caller does not need until its next barrier).  Ends with a barrier when a.wdot is set.
template <int U = 4>
__device__ __forceinline__ void slab_reduce_share(const SlabReduceArgs &a, int wg, int nwg, int tid, int nthreads, double *red) {
    if (a.nsplit == 0) return;                                               // (kernel-uniform)
    const long total = a.n4 + a.k4;
    long per = (total + nwg - 1) / nwg;
    per = (per + 63) & ~63L;                                                 // whole 1 KB runs per workgroup
    const double acc = slab_reduce_range<U>(a, (long)wg * per, (long)wg * per + per, tid, nthreads);
    if (a.wdot) {
        const double s = wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
            for (int i = 0; i < nthreads / 64; ++i) t += red[i];
            if ((long)wg * per < a.n4) atomicAdd(a.dot, (float)t);
        }
        __syncthreads();
    }
}

// ---- host side (core.hip): the reduction a weight-gradient launch left for the next one
bool wgrad_deferred();                                                      // mmdgan_wgrad_defer(1) on the current handle
// scratch for the slabs of a weight-gradient launch on `st` (need bytes).  With deferral on and a reduction pending on the
// same stream, *prev receives it (the caller runs it in its prologue) and the returned region does not overlap its slabs;
// otherwise anything pending is issued as a stand-alone pass first and *prev is empty.  nullptr: no workspace.
void *wgrad_slabs_acquire(size_t need, hipStream_t st, SlabReduceArgs *prev);
// what the launch on `st` leaves behind: with deferral on it becomes the pending reduction, otherwise it is issued right away
int wgrad_slabs_release(const SlabReduceArgs &mine, hipStream_t st);
int wgrad_flush_pending();                                                  // stand-alone pass for whatever is pending (0: ok)
void slab_reduce_launch(const SlabReduceArgs &a, hipStream_t st);           // conv_wino2.hip

}  // namespace mmdgan
