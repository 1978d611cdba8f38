/* ctag_gather.h -- C ABI of the multi-GPU step (libctag_hip.so): frame shards and the final gather of marker lists.
 *
 * The reference is a single-process program (no collective anywhere); north_star adds the
 * only exchange there is: "a batch of independent frames shards trivially across the 8 GPUs of one node with RCCL
 * over xGMI only for the final gather of detected marker lists" (SURVEY.md 8(e)).  One process per GPU, one
 * ctag_handle each; frames are split into contiguous ranges (ctag_shard_range) and every rank ends up with the
 * result records of ALL frames in frame order, byte-identical to a one-GPU run.
 *
 * What travels is not the fixed 11 616-byte record per frame but a packed shard ("counts + used
 * records"): per frame the 16-byte record head (status, n_markers, n_features, flags) followed by its n_markers marker
 * records and n_features feature records.  Exchange = ncclAllGather of the packed sizes, then ONE ncclAllGather of the
 * packed shards padded to the largest, then an unpack kernel that rebuilds the fixed records.
 *
 * RCCL is bound at run time (dlopen of the librccl.so.1 the process already holds, e.g. the one PyTorch loaded, else
 * the system one; CTAG_RCCL_LIB overrides): the library has no link-time dependency on it and single-GPU users never
 * load it.  A C++ host (the reference is C++, main.cpp:44-60) needs nothing but this header; INTEGRATION.md shows
 * the loop.
 */
#ifndef CTAG_GATHER_H
#define CTAG_GATHER_H
#include <stddef.h>
#include <stdint.h>

#include "ctag.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Contiguous frame range [*lo, *hi) of `rank` when n_total frames are split over `world` ranks; the first
 * n_total % world ranks own one frame more.  Pure function, no GPU. */
int ctag_shard_range(int n_total, int rank, int world, int* lo, int* hi);

/* ---- packed shards ------------------------------------------------------------------------------------
 * Layout of a packed shard of n frames (all little-endian, 4-byte aligned):
 *   int32 n_frames, int32 reserved(0), int64 total_bytes                                   16 B
 *   n_frames x { int32 status, n_markers, n_features; uint32 flags }                        16 B each
 *   per frame, in order: its n_markers marker records (ctag_marker_rec, 16 B each), then its n_features feature records
 *   (ctag_feature_rec, 100 B each)
 * Records of frames whose status is not CTAG_OK carry no payload (n_markers = n_features = 0 there). */
size_t ctag_packed_capacity(int n_frames); /* worst-case bytes of a packed shard */
/* device records -> packed shard in device memory (capacity >= ctag_packed_capacity(n)); enqueued on the handle's
 * stream.  If packed_bytes_host is not NULL the call waits and stores the shard's size there. */
int ctag_pack_results(ctag_handle* h, const ctag_frame_result* results_dev, int n, void* packed_dev, size_t capacity,
                      uint64_t* packed_bytes_host);
/* packed shard -> n fixed records in device memory (bytes a record does not use are zero, as the detector writes them) */
int ctag_unpack_results(ctag_handle* h, const void* packed_dev, int n, ctag_frame_result* out_dev);

/* ---- communicator --------------------------------------------------------------------------------------
 * ctag_comm_unique_id: ncclGetUniqueId (rank 0 calls it and hands the 128 bytes to the other ranks by any means).
 * ctag_comm_init: ncclCommInitRank on the handle's device; collective over all ranks.
 * ctag_comm_attach: use a communicator the caller owns (ncclComm_t passed as void*); it is not destroyed here. */
#define CTAG_COMM_ID_BYTES 128
int ctag_comm_unique_id(void* id_bytes);
int ctag_comm_init(ctag_handle* h, const void* id_bytes, int rank, int world);
int ctag_comm_attach(ctag_handle* h, void* nccl_comm, int rank, int world);
int ctag_comm_destroy(ctag_handle* h);
/* the handle's communicator as an ncclComm_t (NULL if none).  A second handle of the same process (its own stream and
 * workspace, e.g. to overlap consecutive batches) gathers through the SAME communicator when it is given to
 * ctag_comm_attach: collectives of one communicator execute in issue order whatever stream they are enqueued on, so a
 * process never holds two communicators whose kernels could start in different orders on different ranks. */
void* ctag_comm_native(ctag_handle* h);
/* text of the last failure of the gather layer on this handle (RCCL / dlopen message), "" if none */
const char* ctag_comm_last_error(ctag_handle* h);

/* ---- the gather ------------------------------------------------------------------------------------------
 * local_dev: this rank's n_local = hi - lo records (ctag_shard_range(n_total, rank, world)), device memory, produced on
 * the handle's stream.  out_dev: n_total records, device memory, identical on every rank afterwards.
 *
 * Two-phase form, so that the host never idles the GPU: _begin enqueues (on the handle's gather stream, ordered after
 * the work already enqueued on the handle's main stream) pack + all-gather of the sizes + their download; the caller
 * may now enqueue the NEXT batch's detection; _end waits for the sizes only, then enqueues the payload all-gather and
 * the unpack and returns; _wait blocks until out_dev is complete.  ctag_gather = the three in a row.
 * One gather may be in flight per handle.
 * Frames that wait for the any-frame workspace (CTAG_PENDING records of a device-memory call, include/ctag.h): _begin does not wait for
 * the detection ahead of it to find out whether there are any -- every rank's count travels with its size, and _end, on EVERY rank, completes
 * them and packs again when some rank has one (rare: cluttered frames).  local_dev must therefore hold its records, unchanged, until _end
 * has returned -- a handle's next batch goes to another buffer (two alternating ones do). */
int ctag_gather_begin(ctag_handle* h, const ctag_frame_result* local_dev, int n_local, int n_total);
int ctag_gather_end(ctag_handle* h, ctag_frame_result* out_dev);
int ctag_gather_wait(ctag_handle* h);
int ctag_gather(ctag_handle* h, const ctag_frame_result* local_dev, int n_local, int n_total, ctag_frame_result* out_dev);
/* Deadline of the exchange's host waits.  A peer that died, or never reaches its collective, must not hold the other ranks for
 * ever: every host wait of the gather (_end's wait for the sizes, _wait, ctag_comm_destroy / ctag_destroy behind a collective in
 * flight) polls with a deadline and asks RCCL for asynchronous errors of the communicator (ncclCommGetAsyncError) while it
 * does.  At the deadline, or on such an error, the communicator is aborted (ncclCommAbort: the collectives in flight end), the call
 * returns CTAG_ERR_HIP and ctag_comm_last_error says why; every handle that shares the communicator fails its later gather calls the
 * same way and the process is expected to exit non-zero.  The abort runs on a thread of the library's (half a second is waited for it at
 * the deadline, two more when the last handle lets go of the communicator); a process whose abort is still busy after that -- RCCL waiting
 * for work ahead of a collective that never started -- must leave through _exit(), not exit(): HIP / RCCL static teardown under a running
 * abort is not safe.  timeout_ms > 0: that many milliseconds; 0: no deadline; < 0: back to the
 * default = the environment variable CTAG_GATHER_TIMEOUT_MS, else 60 000 ms.  Without a communicator (one rank) nothing is bounded:
 * there is no peer to wait for. */
int ctag_gather_set_timeout(ctag_handle* h, int timeout_ms);
/* bytes this rank contributed / the padded per-rank width of the last payload all-gather (introspection for the bench) */
int ctag_gather_last_bytes(ctag_handle* h, uint64_t* local_bytes, uint64_t* padded_bytes);

#ifdef __cplusplus
}
#endif
#endif
