// smcsmc_amd/csrc/pf_probe.hip -- measurement probe, not part of the filter: what does it cost to hand a row over from the
// extend workgroups of row s to those of row s + 1?
//
// The row pipeline pays a kernel boundary per row (pf_hip.hip, k_sweep): the 157 wavefronts of C3's 40 extend workgroups leave
// five partial sums each, the launch ends, the next launch's workgroups read the 785 numbers and redo the row's decision.  The
// alternative is a grid that stays resident: every wavefront publishes its partials with release stores, bumps an arrival
// counter, polls it, then reads the others' partials with coherent loads.  DESIGN.md priced that hand-off from the guide's
// table in round 3; pf_probe_handoff measures both forms on the device, with the same reduction of the same 785 numbers and a
// chosen amount of work per row in between, so that the figure that decides the question is a measurement.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace {

constexpr int NPART = 5;

__device__ __forceinline__ void spin_100mhz(long long ticks) {
    if (ticks <= 0) return;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

// the reduction every workgroup redoes from all wavefronts' partials (decide_row's shape: three levels of a radix-64 tree)
__device__ __forceinline__ double reduce_partials(const double* part, int nw, bool coherent) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int k = lane; k < nw * NPART; k += 64) {
        const double v = coherent ? __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[k];
        acc += v;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
    return acc;
}

// one launch per row: the partials of the previous row are in memory (kernel boundary), this row's are written at the end
__global__ __launch_bounds__(256) void k_handoff_launch(const double* prev, double* next, int nw, long long spin, double* out, int row) {
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    double acc = reduce_partials(prev, nw, false);
    spin_100mhz(spin);
    if (gw < nw && lane < NPART) next[gw * NPART + lane] = acc * 1e-3 + (double)(gw + lane + row);
    if (gw < nw && lane == 0) out[gw] = acc;
}

// resident grid: rows in a loop, hand-off through an arrival counter per row
__global__ __launch_bounds__(256) void k_handoff_resident(double* part /* [2][nw * NPART] */, unsigned* arrive /* [rows] */, int nw, int rows,
                                                          long long spin, double* out, int* err) {
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int r = 0; r < rows; ++r) {
        double* mine = part + (size_t)(r & 1) * nw * NPART;
        spin_100mhz(spin);
        if (gw < nw && lane < NPART) __hip_atomic_store(&mine[gw * NPART + lane], acc * 1e-3 + (double)(gw + lane + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gw < nw) {
            if (lane == 0) {
                __hip_atomic_fetch_add(&arrive[r], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // after this wavefront's partials
                int guard = 0;
                while (__hip_atomic_load(&arrive[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nw) {
                    if (++guard > 4000000) { *err = 1; break; }       // a grid that is not resident as a whole must not hang the box
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;       // somebody gave up waiting: everybody leaves
        acc = reduce_partials(mine, nw, true);
    }
    if (gw < nw && lane == 0) out[gw] = acc;
}

// the same with the hand-off done per workgroup: one arrival and one poller per workgroup (the other wavefronts wait at the
// workgroup barrier), the partials fetched once per workgroup into LDS
__global__ __launch_bounds__(256) void k_handoff_resident_wg(double* part, unsigned* arrive, int nw, int rows, long long spin, double* out, int* err) {
    __shared__ double s_part[4096];
    __shared__ int s_bad;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nwg = (int)gridDim.x, np = nw * NPART;
    double acc = 0.0;
    if (threadIdx.x == 0) s_bad = 0;
    for (int r = 0; r < rows; ++r) {
        double* mine = part + (size_t)(r & 1) * np;
        spin_100mhz(spin);
        if (gw < nw && lane < NPART) __hip_atomic_store(&mine[gw * NPART + lane], acc * 1e-3 + (double)(gw + lane + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&arrive[r], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int guard = 0;
            while (__hip_atomic_load(&arrive[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
                if (++guard > 4000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) s_bad = 1;
        }
        __syncthreads();
        if (s_bad) break;
        for (int k = threadIdx.x; k < np && k < 4096; k += 256) s_part[k] = __hip_atomic_load(&mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        double a2 = 0.0;
        for (int k = lane; k < np && k < 4096; k += 64) a2 += s_part[k];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) a2 += __shfl_xor(a2, m, 64);
        acc = a2;
        __syncthreads();
    }
    if (gw < nw && lane == 0) out[gw] = acc;
}

}  // namespace

// mode 0: `rows` launches of k_handoff_launch back to back on one stream; mode 1: one launch of the resident grid, every wavefront
// arriving and polling for itself; mode 2: the resident grid with one arrival and one poller per workgroup.
// nw wavefronts in workgroups of four; spin_ticks of 10 ns of stand-in work per wavefront and row.  Returns microseconds per row
// (HIP events around the whole sequence) in *us_per_row and a checksum (the two modes compute the same numbers).
extern "C" int pf_probe_handoff(int32_t mode, int32_t rows, int32_t nw, int64_t spin_ticks, double* us_per_row, double* checksum, int32_t device) {
    if (rows < 1 || nw < 1 || nw > 4096 || (mode == 2 && nw * NPART > 4096)) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    const int nwg = (nw + 3) / 4;
    double *part = nullptr, *out = nullptr;
    unsigned* arrive = nullptr;
    int* err = nullptr;
    hipStream_t st;
    hipEvent_t e0, e1;
    if (hipMalloc(&part, (size_t)2 * nw * NPART * 8) != hipSuccess || hipMalloc(&out, (size_t)nw * 8) != hipSuccess ||
        hipMalloc(&arrive, (size_t)rows * 4) != hipSuccess || hipMalloc(&err, 4) != hipSuccess) return -1;
    hipMemset(part, 0, (size_t)2 * nw * NPART * 8); hipMemset(arrive, 0, (size_t)rows * 4); hipMemset(err, 0, 4); hipMemset(out, 0, (size_t)nw * 8);
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    // warm-up launch of either kernel (code object load), outside the timed region
    if (mode == 0) hipLaunchKernelGGL(k_handoff_launch, dim3(nwg), dim3(256), 0, st, part, part + (size_t)nw * NPART, nw, 0LL, out, -1);
    hipStreamSynchronize(st);
    hipMemset(part, 0, (size_t)2 * nw * NPART * 8);
    hipEventRecord(e0, st);
    if (mode == 0) {
        for (int r = 0; r < rows; ++r) {
            double* prev = part + (size_t)((r + 1) & 1) * nw * NPART;
            double* next = part + (size_t)(r & 1) * nw * NPART;
            hipLaunchKernelGGL(k_handoff_launch, dim3(nwg), dim3(256), 0, st, prev, next, nw, (long long)spin_ticks, out, r);
        }
    } else if (mode == 1) {
        hipLaunchKernelGGL(k_handoff_resident, dim3(nwg), dim3(256), 0, st, part, arrive, nw, rows, (long long)spin_ticks, out, err);
    } else {
        hipLaunchKernelGGL(k_handoff_resident_wg, dim3(nwg), dim3(256), 0, st, part, arrive, nw, rows, (long long)spin_ticks, out, err);
    }
    hipEventRecord(e1, st);
    if (mode == 0)          // (untimed) the reduction of the last row's partials, which the resident grid has already done
        hipLaunchKernelGGL(k_handoff_launch, dim3(nwg), dim3(256), 0, st, part + (size_t)((rows + 1) & 1) * nw * NPART, part + (size_t)(rows & 1) * nw * NPART, nw, 0LL, out, rows);
    const hipError_t rc = hipStreamSynchronize(st);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> h(nw);
    int herr = 0;
    hipMemcpy(h.data(), out, (size_t)nw * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    double cs = 0.0;
    for (double v : h) cs += v;
    if (us_per_row) *us_per_row = 1e3 * (double)ms / rows;
    if (checksum) *checksum = cs;
    hipFree(part); hipFree(out); hipFree(arrive); hipFree(err);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(st);
    return (rc == hipSuccess && herr == 0) ? 0 : -2;
}
