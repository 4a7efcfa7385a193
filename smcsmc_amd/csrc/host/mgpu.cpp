// smcsmc_amd/csrc/host/mgpu.cpp -- see mgpu.hpp.  Host code only; compiled with hipcc for the HIP runtime and RCCL
// headers.  One communicator per rank thread (ncclCommInitAll), one stream per rank; the all-gather moves
// ranks * doubles_per_rank * 8 bytes (a few KB): latency-bound, one call per EM iteration.
#include "mgpu.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <stdexcept>

namespace {
void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nccl_ok(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

int visible_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

struct CountAllGather::Impl {
    int ranks = 0;
    size_t per_rank = 0;
    std::vector<int> device;
    // rccl
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<double*> d_send, d_recv;
    // host
    std::vector<double> meeting;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, departed = 0;
    long long round = 0;
};

CountAllGather::CountAllGather(int ranks, const std::vector<int>& device_of_rank, size_t doubles_per_rank, const std::string& transport)
    : impl_(new Impl), transport_(transport) {
    Impl& m = *impl_;
    m.ranks = ranks; m.per_rank = doubles_per_rank; m.device = device_of_rank;
    if (transport_ == "rccl") {
        for (int a = 0; a < ranks; ++a)
            for (int b = a + 1; b < ranks; ++b)
                if (m.device[a] == m.device[b]) throw std::runtime_error("-reduce rccl needs one device per rank");
        m.comm.resize(ranks); m.stream.resize(ranks); m.d_send.resize(ranks); m.d_recv.resize(ranks);
        nccl_ok(ncclCommInitAll(m.comm.data(), ranks, m.device.data()), "ncclCommInitAll");
        for (int r = 0; r < ranks; ++r) {
            hip_ok(hipSetDevice(m.device[r]), "hipSetDevice");
            hip_ok(hipStreamCreateWithFlags(&m.stream[r], hipStreamNonBlocking), "hipStreamCreate");
            hip_ok(hipMalloc((void**)&m.d_send[r], doubles_per_rank * sizeof(double)), "hipMalloc");
            hip_ok(hipMalloc((void**)&m.d_recv[r], (size_t)ranks * doubles_per_rank * sizeof(double)), "hipMalloc");
        }
    } else if (transport_ == "host") {
        m.meeting.assign((size_t)ranks * doubles_per_rank, 0.0);
    } else {
        throw std::runtime_error("-reduce must be rccl or host");
    }
}

CountAllGather::~CountAllGather() {
    Impl& m = *impl_;
    for (size_t r = 0; r < m.comm.size(); ++r) {
        (void)hipSetDevice(m.device[r]);
        (void)hipFree(m.d_send[r]); (void)hipFree(m.d_recv[r]);
        (void)hipStreamDestroy(m.stream[r]);
        (void)ncclCommDestroy(m.comm[r]);
    }
    delete impl_;
}

void CountAllGather::all_gather(int rank, const double* mine, double* all) {
    Impl& m = *impl_;
    const size_t n = m.per_rank;
    if (transport_ == "rccl") {
        hip_ok(hipSetDevice(m.device[rank]), "hipSetDevice");
        hip_ok(hipMemcpyAsync(m.d_send[rank], mine, n * sizeof(double), hipMemcpyHostToDevice, m.stream[rank]), "hipMemcpyAsync");
        nccl_ok(ncclAllGather(m.d_send[rank], m.d_recv[rank], n, ncclDouble, m.comm[rank], m.stream[rank]), "ncclAllGather");
        hip_ok(hipMemcpyAsync(all, m.d_recv[rank], (size_t)m.ranks * n * sizeof(double), hipMemcpyDeviceToHost, m.stream[rank]),
               "hipMemcpyAsync");
        hip_ok(hipStreamSynchronize(m.stream[rank]), "hipStreamSynchronize");
        return;
    }
    // host: everybody writes its block, waits until all have, copies the lot, and the last one out re-arms the meeting
    std::unique_lock<std::mutex> lk(m.mu);
    const long long my_round = m.round;
    std::memcpy(&m.meeting[(size_t)rank * n], mine, n * sizeof(double));
    if (++m.arrived == m.ranks) m.cv.notify_all();
    m.cv.wait(lk, [&] { return m.arrived == m.ranks || m.round != my_round; });
    std::memcpy(all, m.meeting.data(), (size_t)m.ranks * n * sizeof(double));
    if (++m.departed == m.ranks) { m.arrived = 0; m.departed = 0; ++m.round; m.cv.notify_all(); }
    else m.cv.wait(lk, [&] { return m.round != my_round; });
}
