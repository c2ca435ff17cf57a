// Force-directed layout step of Graph::postprocess (reference rvaser/rala
// src/graph.cpp:1132-1226): every point is pushed away from every other point of its component
// by k^2 / d^2 and pulled towards its neighbours (graph edges and removed transitive edges) by
// d / k, then moved by t along the normalised sum.  FP64 throughout, in the reference's order
// of operations and with one accumulator per point walked in ascending point order, so the
// result is bit-identical to a sequential evaluation (no FMA contraction: the library is built
// with -ffp-contract=off; sqrt and division are correctly rounded).
//
// O(n^2) per step: one thread per point, the other points streamed through LDS in tiles.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kTile = 256;

__global__ __launch_bounds__(kTile) void layout_step_kernel(uint32_t n, const double* __restrict__ x,
                                                            const double* __restrict__ y, double* __restrict__ x_out,
                                                            double* __restrict__ y_out,
                                                            const uint32_t* __restrict__ adj_off,
                                                            const uint32_t* __restrict__ adj, double k, double t) {
    __shared__ double sx[kTile], sy[kTile];
    const uint32_t i = blockIdx.x * kTile + threadIdx.x;
    const double px = i < n ? x[i] : 0.0, py = i < n ? y[i] : 0.0;
    double ax = 0.0, ay = 0.0;
    for (uint32_t m0 = 0; m0 < n; m0 += kTile) {
        const uint32_t m = m0 + threadIdx.x;
        sx[threadIdx.x] = m < n ? x[m] : 0.0;
        sy[threadIdx.x] = m < n ? y[m] : 0.0;
        __syncthreads();
        const uint32_t cnt = n - m0 < (uint32_t)kTile ? n - m0 : (uint32_t)kTile;
        if (i < n) {
            for (uint32_t j = 0; j < cnt; ++j) {
                if (m0 + j == i) continue;
                const double dx = px - sx[j], dy = py - sy[j];
                double distance = sqrt(dx * dx + dy * dy);
                if (distance < 0.01) distance = 0.01;
                const double s = (k * k) / (distance * distance);
                ax = ax + dx * s;
                ay = ay + dy * s;
            }
        }
        __syncthreads();
    }
    if (i >= n) return;
    for (uint32_t a = adj_off[i]; a < adj_off[i + 1]; ++a) {
        const uint32_t m = adj[a];
        const double mx = m < n ? x[m] : 0.0, my = m < n ? y[m] : 0.0;     // index n: the origin
        const double dx = px - mx, dy = py - my;
        double distance = sqrt(dx * dx + dy * dy);
        if (distance < 0.01) distance = 0.01;
        const double s = -1. * distance / k;
        ax = ax + dx * s;
        ay = ay + dy * s;
    }
    double length = sqrt(ax * ax + ay * ay);
    if (length < 0.01) length = 0.1;                    // sic (graph.cpp:1208-1210)
    const double s = t / length;
    x_out[i] = px + ax * s;
    y_out[i] = py + ay * s;
}

}  // namespace

void launch_layout_step(uint32_t n, const double* x, const double* y, double* x_out, double* y_out,
                        const uint32_t* adj_off, const uint32_t* adj, double k, double t, hipStream_t s) {
    if (n) {
        hipLaunchKernelGGL(layout_step_kernel, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, n, x, y, x_out, y_out,
                           adj_off, adj, k, t);
    }
}

}  // namespace rala_hip
