// smcsmc_amd/csrc/pf_mp_host.h -- host-side entry points of the structured-model kernels (pf_mp.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "pf_types.h"

#define PF_PMAX 4             // populations supported by the HIP path
#define PF_MMAX 96            // migration events kept per local tree

size_t pf_mp_smem_bytes(int n, int E, int P, int mcap);       // LDS-tree kernels
size_t pf_mp_reg_smem_bytes(int E, int P, int mcap);          // register-tree row kernel
int pf_mp_prepare(size_t smem, int mcap);      // raises the dynamic-LDS limit of the kernels; -1 if the state does not fit
void pf_mp_launch_init(const KArgs& A, double initial_position, size_t smem, hipStream_t st);
// lds_tree: use the LDS-tree kernel whatever the sample size (n > 8 always does)
// fuse (register-tree kernel only, pf_mp_can_fuse): complete the previous row (k_resample's part) while loading
void pf_mp_launch_extend(const KArgs& A, long long s, size_t smem, hipStream_t st, bool lds_tree, int fuse);
bool pf_mp_can_fuse(const KArgs& A, bool lds_tree);
// the extend role of the row pipeline for structured models (k_sweep_xmp, one launch per step; `done` is recorded when it ends)
struct SweepChunk;
size_t pf_mp_sweep_smem_bytes(int E, int P, int mcap, int nc);
int pf_mp_sweep_prepare(size_t smem);
void pf_mp_launch_sweep_x(const KArgs& A, const SweepChunk* tab, long long t, size_t smem, hipStream_t st, hipEvent_t done);
void pf_mp_launch_calibrate(const KArgs& A, unsigned long long seed, long long rep0, long long nrep, int* out_epoch,
                            double* out_dist, int* out_err, size_t smem, hipStream_t st);
void pf_mp_launch_tbl(const KArgs& A, unsigned long long seed, long long nrep, double* out_h, double* out_len, int* out_err,
                      size_t smem, hipStream_t st);
void pf_mp_launch_simulate(const KArgs& A, unsigned long long seed, int nchunks, long long max_sites, double* pos, unsigned* masks,
                           long long* n_sites, int* out_err, size_t smem, hipStream_t st);
