/*
 * frog_comm.h -- C ABI of libfrog_comm.so: the collectives of a multi-GPU registration, for hosts that
 * drive several contexts of libfrog_hip.so from ONE process (one host thread per GPU).
 *
 * The reference parallelises every loop of ImageGroup over images with OpenMP
 * (registration/imageGroup.cxx:239, :572, :912, :1067) inside one address space; here each GPU owns a
 * contiguous range of images (frog_create image_begin / image_end) and the four places where a loop
 * reads ANOTHER image's state become collectives over RCCL (xGMI):
 *
 *   transformPoints  (imageGroup.cxx:910-916)   xyz2 of every image is read by the link loops of all
 *                                               others            -> all-gather of the owned xyz2 rows
 *   updateStats      (:569-598)                 (c1, c2, ratio) of the partner image  -> all-reduce(sum)
 *                                               of the table, rows of other ranks zero
 *   update{Linear,Deformable}Transforms         omp reduction(+) of sDistances / sWeights (:239, :1067)
 *                                               -> all-reduce(sum) of FROG_BUF_ENERGY
 *   updateDeformableTransforms phase B          mean of the proposals over ALL images (:400-432): the
 *                                               shared common-space grid -> all-reduce(sum) of
 *                                               FROG_BUF_GRIDSUM, then of the oversize count (ENERGY[2])
 *   setupDeformableTransforms (:159-179)        bounding box over all images -> all-reduce(min / max)
 *
 * Every rank calls the same function at the same point of the schedule, from its own host thread
 * (hipSetDevice is per thread); calls are enqueued on the context's stream, nothing blocks the host
 * except frog_comm_all_reduce_bounds (it returns host values).  Kept out of libfrog_hip.so so that a host
 * that brings its own collectives (e.g. torch.distributed, frog_amd/distributed.py) never loads RCCL twice.
 */
#ifndef FROG_COMM_H
#define FROG_COMM_H

#include "frog_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct frog_comm frog_comm;

/* One communicator per rank, for the devices devices[0..n_ranks) (ncclCommInitAll: all n created by ONE
 * call, from one thread, before the per-rank threads start).  out[r] belongs to rank r. */
int frog_comm_create_rccl(int n_ranks, const int *devices, frog_comm **out);
/* The same interface without RCCL, for n_ranks contexts that may share one device: the collectives go
 * through a host staging area and barriers between the ranks' threads.  For tests and for rehearsing the
 * multi-rank control flow on a single GPU; not a fast path. */
int frog_comm_create_loopback(int n_ranks, frog_comm **out);
/* One process per GPU (torch.distributed.run, MPI): rank 0 obtains an id (frog_comm_unique_id: ncclGetUniqueId, 128
 * bytes), the launcher's own channel carries it to the other ranks, then every process creates ITS communicator
 * (ncclCommInitRank on `device`).  The collectives below then work exactly as in the one-process form; the xyz2 row
 * range of every rank, which one process reads from its contexts, has to be told: frog_comm_set_rows. */
int frog_comm_unique_id(unsigned char id_out[128]);
int frog_comm_create_rank(int n_ranks, int rank, const unsigned char id[128], int device, frog_comm **out);
/* The same one-process-per-rank interface WITHOUT RCCL: the collectives are staged through POSIX shared memory on the host
 * (named `name`, which every rank of the run passes; rank 0 creates it, nothing is left in /dev/shm afterwards) and a
 * barrier between the ranks' processes that gives up after two minutes.  Ranks may share a device.  What it is for: rehearsing the
 * one-process-per-GPU host on a box with a single GPU, and the fallback that still yields a (host-staged) multi-GPU run
 * when RCCL itself does not come up.  frog_comm_bind / frog_comm_set_rows / the collectives work as for frog_comm_create_rank. */
int frog_comm_create_shm(int n_ranks, int rank, const char *name, int device, frog_comm **out);
/* row_begin[r] .. row_begin[r + 1] = the xyz2 rows (points) of rank r; n_ranks + 1 entries.  Call after frog_comm_bind. */
int frog_comm_set_rows(frog_comm *comm, const uint64_t *row_begin);
/* Destroys all n communicators of one create call (pass the array it filled; n = 1 for frog_comm_create_rank). */
void frog_comm_destroy_all(int n_ranks, frog_comm **comms);

/* `ctx` must be the context of this communicator's rank; image_begin[r] .. image_begin[r + 1] are the
 * images of rank r (n_ranks + 1 entries, the same on every rank). */
int frog_comm_bind(frog_comm *comm, frog_ctx *ctx, const uint32_t *image_begin);

/* all-gather of the owned rows of FROG_BUF_XYZ2 into every rank's replica (RCCL: the rows go through the slab below --
 * one equal-size all-gather of padded slots + one unpack launch; until round 5 n grouped broadcasts) */
int frog_comm_all_gather_xyz2(frog_comm *comm);
/* ---- the padded coordinate gather (include/frog_hip.h, "Two collectives per deformable iteration") -----------------
 * The communicator owns the slab (on its rank's device; slot_rows = the longest shard).
 *   frog_comm_slab             its address and slot length (allocated on the first call, once every rank's rows are known)
 *   frog_comm_all_gather_slab  ONE equal-size all-gather of the ranks' slots, in place (ncclAllGather; host-staged on the
 *                              shared-memory and loopback transports)
 *   frog_comm_gather_points    transformPoints of the rank's images + the gather + every slot into FROG_BUF_XYZ2:
 *                              frog_transform_points_slab, frog_comm_all_gather_slab, frog_comm_unpack_slab_step(sum_mask).
 *                              With sum_mask != 0 the step's scalars ride along: call frog_step_finish(ctx, &E) next. */
int frog_comm_slab(frog_comm *comm, void **slab, uint64_t *slot_rows);
int frog_comm_all_gather_slab(frog_comm *comm);
int frog_comm_gather_points(frog_comm *comm, int apply, int after_step, uint32_t sum_mask);

/* all-reduce(sum) of FROG_BUF_EM (float), FROG_BUF_ENERGY or FROG_BUF_GRIDSUM (double), in place */
int frog_comm_all_reduce(frog_comm *comm, int which);
/* group-wide box from the ranks' own boxes (frog_bounds_local): mins / maxs are replaced */
int frog_comm_all_reduce_bounds(frog_comm *comm, double mins[3], double maxs[3]);
/* host-side barrier between the ranks' threads (both kinds) */
int frog_comm_barrier(frog_comm *comm);
/* Device time of the collectives: with on != 0 every eighth call of a kind is bracketed by a pair of HIP events on the
 * context's stream.  frog_comm_timing_read waits for the stream and returns, per kind (0 all_gather_xyz2, 1 all_reduce of
 * FROG_BUF_EM, 2 of FROG_BUF_ENERGY, 3 of FROG_BUF_GRIDSUM): the milliseconds of the sampled calls, all calls, sampled calls. */
int frog_comm_timing(frog_comm *comm, int on);
int frog_comm_timing_read(frog_comm *comm, double ms[4], uint64_t calls[4], uint64_t sampled[4]);

#ifdef __cplusplus
}
#endif
#endif /* FROG_COMM_H */
