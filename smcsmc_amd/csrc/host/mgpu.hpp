// smcsmc_amd/csrc/host/mgpu.hpp -- the one exchange step of a multi-chunk E-step: every rank contributes the packed
// sufficient statistics (CountModel) of its chunks, every rank receives all of them.  The functional ancestor is the
// file-based sum of the front-end (smcsmc/model.py:1176-1184); here it is one RCCL all-gather over xGMI, called
// directly from the host threads that drive the devices, followed by a sum in chunk order on every rank (so the
// result does not depend on the number of ranks or on arrival order).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

class CountAllGather {
  public:
    // ranks of this process, the device each of them drives, doubles contributed per rank.
    // transport "rccl": ncclAllGather between the ranks' devices (needs one device per rank);
    // transport "host": the ranks' blocks meet in host memory (ranks that share a device, or no RCCL wanted).
    CountAllGather(int ranks, const std::vector<int>& device_of_rank, size_t doubles_per_rank, const std::string& transport);
    ~CountAllGather();
    CountAllGather(const CountAllGather&) = delete;
    // called by every rank thread, concurrently; `all` receives ranks * doubles_per_rank doubles in rank order
    void all_gather(int rank, const double* mine, double* all);
    const std::string& transport() const { return transport_; }

  private:
    struct Impl;
    Impl* impl_;
    std::string transport_;
};

int visible_devices();
