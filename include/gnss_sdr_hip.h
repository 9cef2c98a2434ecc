/*
 * gnss_sdr_hip.h -- C ABI of the MI355X (gfx950) acquisition + tracking correlator engine.
 *
 * This is the drop-in boundary for gnss-sdr's one data-parallel hot path.  Plain C,
 * plain pointers and sizes, opaque handles, int status returns (0 = GSH_OK); no HIP,
 * torch or C++ types cross it and no exception ever does.  The shared library that
 * implements it is gnss-sdr_amd/libgnss_sdr_hip.so (hand-written HIP, built by
 * __graft_entry__.build()).  There is NO CPU fallback: every entry point that needs the
 * GPU returns GSH_ERR_NO_DEVICE / GSH_ERR_HIP when it cannot run there.
 *
 * Each group of entry points names the reference interface it replaces; paths are
 * relative to the gnss-sdr tree (/root/reference):
 *   mcorr.h  = src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h
 *   mcorr.cc = src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.cc
 *   acq.cc   = src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc
 *   acq.h    = src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.h
 *   trk.cc   = src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc
 *
 * Complex samples are interleaved little-endian float32 I,Q everywhere ("iq" pointers
 * address 2*n floats), i.e. gr_complex / std::complex<float> / lv_32fc_t memory.
 */
#ifndef GNSS_SDR_HIP_H
#define GNSS_SDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define GSH_ABI_VERSION 25
#define GSH_MAX_TAPS 8 /* VE/E/P/L/VL needs 5 (trk.cc:609-650); 8 leaves room for multi-tap dumps */

    enum
    {
        GSH_OK = 0,
        GSH_ERR_INVALID = 1,   /* bad argument (null pointer, size out of range, taps > GSH_MAX_TAPS ...) */
        GSH_ERR_NO_DEVICE = 2, /* no HIP device / device index out of range */
        GSH_ERR_HIP = 3,       /* a HIP runtime call failed; gsh_last_error() has the text */
        GSH_ERR_STATE = 4,     /* call order violated (e.g. correlate before set_local_code) */
        GSH_ERR_UNSUPPORTED = 5
    };

    /* ------------------------------------------------------------------ library */
    int gsh_abi_version(void);
    int gsh_device_count(void);               /* 0 when no GPU is visible */
    const char* gsh_last_error(void);         /* thread-local, never NULL */
    int gsh_device_name(int device, char* buf, size_t buflen);
    /* diagnostic: streaming-read bandwidth of the device as THIS library's loads see it -- a `bytes`-sized buffer (allocated here; larger than the 256 MiB Infinity
     * Cache to measure HBM) read `reps` times with 16-byte loads by a full grid, HIP-event timed; *gb_per_s = bytes x reps / time.  The figure rooflines are
     * stated against beside the nominal peak. */
    int gsh_probe_read_bandwidth(int device, uint64_t bytes, int reps, double* gb_per_s);

    /* ================================================================== TRACKING
     * (1) gsh_mcorr_*: one-to-one replacement of class Cpu_Multicorrelator_Real_Codes
     *     (mcorr.h:37-61).  Same call order, same argument meaning, same borrowed-pointer
     *     rules as the reference: `shifts_chips`, `corr_out` and `sig_in` are BORROWED
     *     (read / written at every correlate call, mcorr.cc:53-72); the local code is
     *     copied to the device once per set_local_code_and_taps (the reference keeps a
     *     pointer, trk.cc:1030 sets it once per start_tracking).
     *     The reference's bool returns (always true, never checked: mcorr.cc:49,62,71,125)
     *     become int status codes.
     */
    typedef struct gsh_mcorr gsh_mcorr_t;

    int gsh_mcorr_create(int device, gsh_mcorr_t** out);
    void gsh_mcorr_destroy(gsh_mcorr_t* h);

    /* mcorr.cc:36-50  bool init(int max_signal_length_samples, int n_correlators) */
    int gsh_mcorr_init(gsh_mcorr_t* h, int max_signal_length_samples, int n_correlators);
    /* mcorr.cc:53-63  bool set_local_code_and_taps(int, const float*, float*) */
    int gsh_mcorr_set_local_code_and_taps(gsh_mcorr_t* h, int code_length_chips, const float* local_code_in, float* shifts_chips);
    /* mcorr.cc:66-72  bool set_input_output_vectors(std::complex<float>*, const std::complex<float>*) */
    int gsh_mcorr_set_input_output_vectors(gsh_mcorr_t* h, float* corr_out_iq, const float* sig_in_iq);
    /* mcorr.cc:163-167 void set_high_dynamics_resampler(bool)   (object default: true, mcorr.h:60) */
    int gsh_mcorr_set_high_dynamics_resampler(gsh_mcorr_t* h, int use_high_dynamics_resampler);
    /* mcorr.cc:103-126 bool Carrier_wipeoff_multicorrelator_resampler(7 args): synchronous,
     * corr_out[0..n_correlators) is valid on return. */
    int gsh_mcorr_carrier_wipeoff_multicorrelator_resampler(gsh_mcorr_t* h, float rem_carrier_phase_in_rad,
        float phase_step_rad, float phase_rate_step_rad, float rem_code_phase_chips, float code_phase_step_chips,
        float code_phase_rate_step_chips, int signal_length_samples);
    /* mcorr.cc:129-144 6-argument overload: always the standard rotator, but the code
     * resampler still follows the high-dynamics flag (mcorr.cc:137 calls update_local_code). */
    int gsh_mcorr_carrier_wipeoff_multicorrelator_resampler6(gsh_mcorr_t* h, float rem_carrier_phase_in_rad,
        float phase_step_rad, float rem_code_phase_chips, float code_phase_step_chips,
        float code_phase_rate_step_chips, int signal_length_samples);
    /* mcorr.cc:147-160 bool free() */
    int gsh_mcorr_free(gsh_mcorr_t* h);

    /*
     * (2) gsh_bank_*: the batched form the GPU wants -- one launch serves every
     *     (channel, epoch) "job" that is ready.  A job is exactly one reference
     *     Carrier_wipeoff_multicorrelator_resampler call (mcorr.cc:103-126) whose input
     *     window is addressed by sample index inside a device-resident IF stream instead
     *     of a host pointer, so each sample crosses PCIe / xGMI once, not once per channel.
     */
    typedef struct gsh_bank gsh_bank_t;

    typedef struct gsh_corr_job
    {
        uint64_t sample_offset;      /* first sample of the window inside the attached stream */
        int32_t n_samples;           /* signal_length_samples, trk.cc:1243 passes vector_length */
        int32_t code_slot;           /* which uploaded local code (one per channel / PRN) */
        float rem_carr_phase_rad;    /* mcorr.cc:104 */
        float phase_step_rad;        /* mcorr.cc:105 */
        float phase_rate_step_rad;   /* mcorr.cc:106 (used only when high_dyn) */
        float rem_code_phase_chips;  /* mcorr.cc:107, already multiplied by samples-per-chip (trk.cc:1240) */
        float code_phase_step_chips; /* mcorr.cc:108 */
        float code_phase_rate_step_chips; /* mcorr.cc:109 (used only when high_dyn) */
        int32_t n_taps;              /* 1..GSH_MAX_TAPS */
        int32_t high_dyn;            /* 0: resampler_32f_xn + rotator_dot_prod; 1: the high-dynamics pair */
        float shifts_chips[GSH_MAX_TAPS]; /* tap offsets in code samples, ascending (trk.cc:632-648) */
    } gsh_corr_job;                  /* 80 bytes, POD */

    int gsh_bank_create(int device, int n_code_slots, int max_code_length, gsh_bank_t** out);
    void gsh_bank_destroy(gsh_bank_t* b);
    /* copy one real-valued local code (trk.cc:810-1028 generate it) into slot `slot` */
    int gsh_bank_set_code(gsh_bank_t* b, int slot, const float* code, int code_length);
    /* IF sample stream.  _host copies n samples to the device (H2D);
     * _device borrows device memory the caller keeps alive (16-byte aligned). */
    int gsh_bank_set_stream_host(gsh_bank_t* b, const float* iq, uint64_t n_samples);
    int gsh_bank_set_stream_device(gsh_bank_t* b, const void* device_iq, uint64_t n_samples);
    /* one synchronous batch: upload jobs, launch, download.  out_iq: n_jobs*GSH_MAX_TAPS complex64
     * (job-major, tap-minor; taps >= n_taps are zero). */
    int gsh_bank_correlate(gsh_bank_t* b, const gsh_corr_job* jobs, int n_jobs, float* out_iq);
    /* the same, split so that a caller can keep the job table and results on the device and
     * time launches alone: upload once, launch many, read once.  hip_stream: a hipStream_t cast to
     * void* (NULL = the bank's own stream).  gsh_bank_launch is asynchronous. */
    /* Pair fusion (default on): a single-tap job placed directly after a job with the same window and the same NCO parameters -- the
     * data-component prompt that track_pilot adds to a pilot channel (trk.cc:1246-1256: a second correlator object over the same
     * samples) -- is computed by that job's work-groups while they hold the rotated samples, instead of a second pass over the window.
     * Results are unchanged (each tap is the same sum over the same products in the same order); 0 disables it for A/B runs. */
    int gsh_bank_set_pair_fusion(gsh_bank_t* b, int enable);
    int gsh_bank_upload_jobs(gsh_bank_t* b, const gsh_corr_job* jobs, int n_jobs);
    int gsh_bank_launch(gsh_bank_t* b, void* hip_stream);
    int gsh_bank_synchronize(gsh_bank_t* b);
    int gsh_bank_read_outputs(gsh_bank_t* b, float* out_iq, int n_jobs);
    /* HIP-event timing of `reps` back-to-back launches of the uploaded job table on the bank's
     * stream: average milliseconds per launch of the correlator kernel. */
    int gsh_bank_time_launches(gsh_bank_t* b, int reps, float* avg_ms);
    /* work-groups per job used by the next launches (1 = throughput mode; >1 spreads one epoch
     * over more CUs for latency-bound closed-loop use).  0 = choose automatically. */
    int gsh_bank_set_splits(gsh_bank_t* b, int splits);
    /* an offset (in samples) added to every job's sample_offset at launch time: a job table uploaded once -- window positions relative to
     * the start of a block -- serves block after block of a long stream or of a ring (positions are taken modulo the ring's capacity)
     * without being re-staged and re-uploaded every time.  The caller keeps the shifted windows inside the stream / resident in the ring. */
    int gsh_bank_set_sample_base(gsh_bank_t* b, uint64_t sample_base);

    /*
     * (3) The 16-bit family: gsh_mcorr16_* replaces class Cpu_Multicorrelator_16sc
     *     (src/algorithms/tracking/libs/cpu_multicorrelator_16sc.h:38-57, .cc:25-120) method for method, gsh_bank16_* is its batched form.
     *     Samples, local code and correlator outputs are complex int16 (lv_16sc_t: interleaved I, Q).  Results are those of the reference's
     *     generic protokernels BIT FOR BIT (K/volk_gnsssdr_16ic_xn_resampler_16ic_xn.h:60-78, K/volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn.h:66-102):
     *     float32 rotation rounded to int16 per sample, the phasor recurrence with its renormalisation every 256 samples, integer products kept
     *     to 16 bits, sums that saturate at every addition in sample order.  (The reference's SSE / AVX2 protokernels of the same kernel differ
     *     from its generic one by a few units -- four interleaved partial sums, another phase schedule; the generic one is what its QA compares
     *     against and what is reproduced here.)  No block of the reference instantiates this class; it is built for completeness of the family.
     */
    typedef struct gsh_mcorr16 gsh_mcorr16_t;
    int gsh_mcorr16_create(int device, gsh_mcorr16_t** out);
    void gsh_mcorr16_destroy(gsh_mcorr16_t* h);
    /* cpu_multicorrelator_16sc.cc:25-41  bool init(int max_signal_length_samples, int n_correlators) */
    int gsh_mcorr16_init(gsh_mcorr16_t* h, int max_signal_length_samples, int n_correlators);
    /* .cc:44-53  bool set_local_code_and_taps(int, const lv_16sc_t*, float*): the code is copied to the device, `shifts_chips` is BORROWED and re-read at every call */
    int gsh_mcorr16_set_local_code_and_taps(gsh_mcorr16_t* h, int code_length_chips, const int16_t* local_code_in_iq, float* shifts_chips);
    /* .cc:56-62  bool set_input_output_vectors(lv_16sc_t* corr_out, const lv_16sc_t* sig_in): both BORROWED */
    int gsh_mcorr16_set_input_output_vectors(gsh_mcorr16_t* h, int16_t* corr_out_iq, const int16_t* sig_in_iq);
    /* .cc:80-96  bool Carrier_wipeoff_multicorrelator_resampler(5 args): synchronous, corr_out[0 .. n_correlators) valid on return */
    int gsh_mcorr16_carrier_wipeoff_multicorrelator_resampler(gsh_mcorr16_t* h, float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips,
        float code_phase_step_chips, int signal_length_samples);
    /* .cc:107-120 bool free() */
    int gsh_mcorr16_free(gsh_mcorr16_t* h);

    typedef struct gsh_bank16 gsh_bank16_t;
    typedef struct gsh_corr16_job
    {
        uint64_t sample_offset;      /* first sample of the window inside the attached int16 stream */
        int32_t n_samples;           /* signal_length_samples */
        int32_t code_slot;
        float rem_carr_phase_rad;    /* .cc:81 */
        float phase_step_rad;        /* .cc:82 */
        float rem_code_phase_chips;  /* .cc:83 */
        float code_phase_step_chips; /* .cc:84 */
        int32_t n_taps;              /* 1..GSH_MAX_TAPS */
        int32_t reserved;
        float shifts_chips[GSH_MAX_TAPS];
    } gsh_corr16_job;                /* 72 bytes, POD */
    int gsh_bank16_create(int device, int n_code_slots, int max_code_length, gsh_bank16_t** out);
    void gsh_bank16_destroy(gsh_bank16_t* b);
    int gsh_bank16_set_code(gsh_bank16_t* b, int slot, const int16_t* code_iq, int code_length);
    /* the IF stream as complex int16: _host copies it to the device, _device borrows device memory (4-byte aligned) */
    int gsh_bank16_set_stream_host(gsh_bank16_t* b, const int16_t* iq, uint64_t n_samples);
    int gsh_bank16_set_stream_device(gsh_bank16_t* b, const void* device_iq, uint64_t n_samples);
    /* one synchronous batch; out_iq: n_jobs * GSH_MAX_TAPS complex int16 (job-major, tap-minor; taps >= n_taps are zero).  The two carrier phasors of
     * every job are formed on the host with the C library, by the expressions of cpu_multicorrelator_16sc.cc:89-93. */
    int gsh_bank16_correlate(gsh_bank16_t* b, const gsh_corr16_job* jobs, int n_jobs, int16_t* out_iq);
    /* the same in pieces (upload once, launch many, read once); gsh_bank16_launch is asynchronous on the bank's stream */
    int gsh_bank16_upload_jobs(gsh_bank16_t* b, const gsh_corr16_job* jobs, int n_jobs);
    int gsh_bank16_launch(gsh_bank16_t* b);
    int gsh_bank16_read_outputs(gsh_bank16_t* b, int16_t* out_iq, int n_jobs);
    /* HIP-event timing of `reps` back-to-back launches (both kernels) of the uploaded job table: average milliseconds per launch */
    int gsh_bank16_time_launches(gsh_bank16_t* b, int reps, float* avg_ms);

    /* ================================================================ SAMPLE STREAM (device-resident ring)
     * gsh_stream_*: the IF sample stream of one RF front-end kept in device memory, addressed by ABSOLUTE sample index
     * (sample 0 = the first sample ever pushed), so that every channel's correlation window refers to bytes that crossed
     * PCIe / xGMI once.  In the reference every tracking / acquisition block reads the same GNU Radio buffer
     * (gnss_flowgraph.cc:1227-1231); this is that buffer's device-side twin.  Raw front-end formats are converted on the
     * device with the arithmetic of the reference's data_type_adapter blocks (ibyte_to_complex.cc:45-51,
     * ishort_to_complex.cc:45-51: integer -> float cast, no scaling; inverted_spectrum -> conjugate).
     * The ring keeps the last `capacity_samples`; any window of up to `max_window_samples` is contiguous in device
     * memory (the first max_window samples are mirrored behind the end), so the correlator kernels never wrap. */
    typedef struct gsh_stream gsh_stream_t;
    enum gsh_item_type
    {
        GSH_ITEM_GR_COMPLEX = 0, /* interleaved float32 I,Q (gr_complex), 8 bytes per sample */
        GSH_ITEM_SHORT = 1,      /* interleaved int16 I,Q (item_type ishort / cshort), 4 bytes per sample */
        GSH_ITEM_BYTE = 2        /* interleaved int8 I,Q (item_type ibyte / cbyte), 2 bytes per sample */
    };
    int gsh_stream_create(int device, uint64_t capacity_samples, uint32_t max_window_samples, gsh_stream_t** out);
    /* Destroying a ring that tracking handles are bound to (gsh_trk_set_stream_ring) tells those handles -- a live residency on it is wound down first, then the
     * memory goes -- but it is NOT synchronised against other threads' calls on those handles or against their destruction: the caller keeps gsh_stream_destroy
     * from racing with gsh_trk_* calls on handles bound to the ring (Hip_Tracking_Runtime does, under its handle mutex). */
    void gsh_stream_destroy(gsh_stream_t* s);
    /* append n samples held in host memory; *first_index (may be NULL) receives the absolute index of items[0].
     * Synchronous: the samples are resident (and any older ones they displace are gone) on return. */
    int gsh_stream_push(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    /* the same with the items already in device memory (e.g. a block received over RCCL); the conversion is queued on
     * `hip_stream` (a hipStream_t cast to void*, NULL = the ring's own stream and synchronous) */
    int gsh_stream_push_device(gsh_stream_t* s, const void* device_items, uint64_t n, int item_type, int inverted_spectrum, void* hip_stream,
        uint64_t* first_index);
    /* [*oldest, *next): the absolute sample indices currently resident */
    /* Asynchronous ingest: the host-to-device copy and the conversion are queued on the ring's own stream and the call returns at once, so
     * the next block travels over PCIe while the correlators (which wait on the ring's event, not on the host) work on the previous one
     * (the reference's flowgraph streams continuously, gnss_flowgraph.cc:1227-1231).  `items` must stay valid and unmodified until
     * gsh_stream_wait() returns or two further pushes have been issued; pinned (page-locked) host memory makes the copy a true DMA.
     * Two device staging buffers alternate.  The caller keeps the ring from lapping its readers: a push may only overwrite samples no
     * queued launch still reads (capacity >= everything in flight). */
    int gsh_stream_push_async(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    int gsh_stream_wait(gsh_stream_t* s);  /* host waits for every queued push */
    /* Asynchronous ingest out of memory the caller wants back at once (a GNU Radio input buffer): the items are copied into page-locked
     * staging memory owned by the ring (four buffers in rotation), from where the copy to the device and the conversion are queued on the
     * ring's stream; `items` may be reused as soon as the call returns.  Readers wait on the ring's events as for every other push. */
    int gsh_stream_push_staged(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    /* The same for items that already lie in page-locked host memory (gsh_host_register, or memory the caller allocated page-locked): no staging
     * copy on the host -- the DMA engine reads `items` directly -- and the call returns when that DMA has finished (the conversion into the ring
     * stays queued), so `items` may be re-used on return.  This is the path for a GNU Radio input buffer: the scheduler re-uses the same buffer
     * for the whole run, so registering it once makes every later push a true DMA. */
    int gsh_stream_push_pinned(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    /* the same in two halves: _async queues the DMA and the conversion and returns at once; gsh_stream_wait_copied blocks until every DMA queued
     * so far has read its source (`items` of every earlier _async call may be re-used then).  wait_copied touches none of the ring's bookkeeping and
     * may be called WITHOUT the serialisation the other entry points need -- a caller that guards the ring with a lock queues under the lock and
     * waits outside it, so that launches which read the ring are queued while the DMA runs. */
    int gsh_stream_push_pinned_async(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    int gsh_stream_wait_copied(gsh_stream_t* s);
    /* ... and for a caller that gets its memory back piece by piece (a GNU Radio block returns input to the scheduler only up to what it consumes): blocks
     * until every push that covers samples below end_index has been read out of the caller's memory (and has reached the ring); pushes queued behind
     * those may still be in flight.  *complete_upto (may be NULL): the index up to which the ring is now known to be complete.  Any thread. */
    int gsh_stream_wait_copied_upto(gsh_stream_t* s, uint64_t end_index, uint64_t* complete_upto);
    /* page-lock / release a range of host memory for DMA (hipHostRegister / hipHostUnregister behind the ABI: host code above it has no HIP headers).
     * The range must be mapped; registering pages twice fails with GSH_ERR_HIP. */
    int gsh_host_register(int device, void* ptr, size_t bytes);
    int gsh_host_unregister(void* ptr);
    /* position an (idle) ring: the next pushed sample gets absolute index next_index and nothing older is resident -- a channel that
     * starts hours into a run does not have to fill the ring from index 0 */
    int gsh_stream_seek(gsh_stream_t* s, uint64_t next_index);
    /* ---- one block, N GPUs: replication of the IF sample block into N device rings over RCCL / xGMI.
     * Channels / PRNs shard over the GPUs of a node; they all read the same stream (gnss_flowgraph.cc:1227-1231), so every block enters ONE
     * GPU -- rank 0, the ingest GPU -- in the front-end's raw item format and is replicated to the others, each converting it to complex64
     * into its own ring.  GSH_GROUP_BROADCAST: ncclBroadcast.  GSH_GROUP_SCATTER_ALLGATHER: rank 0 sends a different 1/N to every peer over
     * all its xGMI links at once, then an all-gather completes the block (xGMI is point to point: every link carries block/N twice instead
     * of one link carrying the whole block per hop).  The rings are ordinary gsh_stream_t: banks, loops and acquisition handles bind to
     * them as usual and wait on their events; nothing here synchronises the host.  A push is queued on the rings' own streams; consecutive
     * pushes alternate between two staging buffers.  Ring capacity must cover what is in flight (see gsh_stream_push_async). */
    typedef struct gsh_stream_group gsh_stream_group_t;
#define GSH_GROUP_BROADCAST 0
#define GSH_GROUP_SCATTER_ALLGATHER 1
/* OR'ed into `mode`: a group of ONE rank normally needs no exchange and never touches RCCL; with this flag it builds its communicator
 * (ncclCommInitAll / ncclCommInitRank with one rank) and sends every block through the mode's collectives all the same -- ncclBroadcast, or
 * the grouped ncclSend / ncclRecv to itself + ncclAllGather -- so that the dlopen, the hand-declared signatures and the stream ordering can be
 * exercised on a single-GPU box.  The environment variable GSH_GROUP_FORCE_RCCL=1 sets it for every group of one.  No effect on world > 1. */
#define GSH_GROUP_FORCE_RCCL 0x100
    /* one process drives all n_devices GPUs (ncclCommInitAll); local ring i belongs to devices[i], rank i */
    int gsh_stream_group_create(const int* devices, int n_devices, uint64_t capacity_samples, uint32_t max_window_samples, int mode, gsh_stream_group_t** out);
    /* one process per GPU: id128 = 128 bytes from gsh_comm_unique_id() on one rank, handed to every rank by whoever launched them
     * (environment, file, MPI, torch.distributed's store ... -- control plane only) */
    int gsh_comm_unique_id(void* id128);
    int gsh_stream_group_create_rank(int device, int rank, int world, const void* id128, uint64_t capacity_samples, uint32_t max_window_samples, int mode,
        gsh_stream_group_t** out);
    void gsh_stream_group_destroy(gsh_stream_group_t* g);
    int gsh_stream_group_size(const gsh_stream_group_t* g);                       /* local rings */
    gsh_stream_t* gsh_stream_group_ring(gsh_stream_group_t* g, int local_index);  /* owned by the group */
    /* every rank calls it for every block (a collective); the process that owns rank 0 supplies the block (host memory, or -- _device --
     * memory on rank 0's GPU), the others pass NULL */
    int gsh_stream_group_push(gsh_stream_group_t* g, const void* host_items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    int gsh_stream_group_push_device(gsh_stream_group_t* g, const void* device_items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index);
    int gsh_stream_group_wait(gsh_stream_group_t* g);
    /* what the group has done with RCCL so far: *rccl_ranks = ranks of the communicator(s) the group created (0: none -- a group of one without
     * GSH_GROUP_FORCE_RCCL), *rccl_version = ncclGetVersion's code (0 when RCCL was never loaded), *collectives = RCCL collective / point-to-point
     * calls this group has issued (broadcasts, sends, receives, all-gathers).  Any pointer may be NULL. */
    int gsh_stream_group_rccl_info(const gsh_stream_group_t* g, int32_t* rccl_ranks, int32_t* rccl_version, uint64_t* collectives);
    /* the exchange plan of ONE rank for one block of `bytes` raw bytes: the list of collective / point-to-point operations gsh_stream_group_push issues
     * for it, in order -- the single place where chunk sizes, offsets and peers are decided (the push walks this very list).  Operations of the same
     * `phase` go out between one ncclGroupStart / ncclGroupEnd.  Buffers: STAGE = the rank's staging buffer of *padded_bytes (the block, zero-extended to
     * world x chunk, chunk a multiple of 16), PIECE = its chunk-sized scratch.  Needs no GPU: a launcher can size its messages from it, the CPU suite
     * executes it over torch.distributed / gloo (tests/test_sharding_gloo.py).  ops may be NULL (count only). */
#define GSH_GROUP_OP_BROADCAST 0 /* ncclBroadcast(src, dst, bytes, root = peer) */
#define GSH_GROUP_OP_SEND 1      /* ncclSend(src, bytes, peer) */
#define GSH_GROUP_OP_RECV 2      /* ncclRecv(dst, bytes, peer) */
#define GSH_GROUP_OP_ALLGATHER 3 /* ncclAllGather(src, dst, bytes per rank) */
#define GSH_GROUP_BUF_NONE 0
#define GSH_GROUP_BUF_STAGE 1
#define GSH_GROUP_BUF_PIECE 2
    typedef struct
    {
        int32_t op;
        int32_t peer;
        int32_t phase;
        int32_t src_buf, dst_buf;
        uint64_t src_offset, dst_offset;
        uint64_t bytes;
    } gsh_group_op_t;
    int gsh_stream_group_plan(uint64_t bytes, int world, int rank, int mode, gsh_group_op_t* ops, int max_ops, int* n_ops, uint64_t* padded_bytes);
    /* the collective library in use: file name of the loaded librccl (the system's, or what the environment variable GSH_RCCL_LIBRARY names -- a site's own
     * build, or the test-only stand-in tests/host/libfake_rccl.so with which the N > 1 paths run on a one-GPU box).  GSH_RCCL_LIBRARY is read whenever a
     * group or an id is made; it must not change while a group made with the other library is alive. */
    int gsh_comm_library(char* path, int capacity);
    int gsh_stream_range(gsh_stream_t* s, uint64_t* oldest, uint64_t* next);
    /* copy resident samples [index, index + n) back to the host as complex64 (tests, dumps) */
    int gsh_stream_read(gsh_stream_t* s, uint64_t index, uint64_t n, float* out_iq);
    /* stand-alone conversion of n device-resident items to complex64 (device_dst 8-byte aligned), asynchronous on hip_stream */
    int gsh_convert_samples_device(int device, const void* device_items, int item_type, int inverted_spectrum, void* device_dst, uint64_t n,
        void* hip_stream);
    /* Direct (nearest-neighbour) resampler, the arithmetic of direct_resampler_conditioner_cc (src/algorithms/resampler/gnuradio_blocks/
     * direct_resampler_conditioner_cc.cc:39-129; used by the signal conditioner and the acquisition decimator, gnss_flowgraph.cc:1165-1209):
     * 32-bit phase accumulator, a sample is copied on every wrap.  Stateless form: outputs are numbered from the start of the stream,
     * output j reads input ceil(j 2^32 / phase_step) (decimation) or floor((j + 1) phase_step / 2^32) (interpolation) -- what the reference's
     * loop produces, however the stream is cut into blocks.  device_src[0] is absolute input sample in0 (n_in samples, complex64), device_dst[0]
     * receives absolute output out0; *n_out = how many outputs this block feeds (<= max_out), *n_in_consumed (may be NULL) = input samples the
     * reference block would have consumed by then.  Asynchronous on hip_stream. */
    int gsh_direct_resample_device(int device, const void* device_src, uint64_t in0, uint64_t n_in, double fs_in, double fs_out, uint64_t out0,
        void* device_dst, uint64_t max_out, uint64_t* n_out, uint64_t* n_in_consumed, void* hip_stream);
    /* Frequency-translating decimating FIR filter: what the input-filter adapters instantiate (src/algorithms/input_filter/adapters/
     * freq_xlating_fir_filter.cc:115-161: gr::filter::freq_xlating_fir_filter_{ccf,fcf,scf}::make(decimation, taps, IF, sampling_frequency);
     * fir_filter.cc: fir_filter_ccf = the same with IF 0, decimation 1).  y[m] = sum_k taps[k] x[mD - k] exp(-j 2 pi IF (mD - k) / fs), zero
     * history before the stream starts.  The taps are the adapter's (pm_remez / firdes, computed once on the host).  input_kind: 0 complex64
     * (ccf), 1 real float32 (fcf), 2 real int16, 3 real int8 (scf after byte_to_short).  The handle carries the filter history, so a stream may
     * be fed in blocks of any size: *n_out outputs are written per call (every output whose newest input has arrived). */
    typedef struct gsh_fir gsh_fir_t;
    int gsh_fir_create(int device, const float* taps, int n_taps, int decimation, double center_freq_hz, double sampling_freq_hz, int input_kind,
        gsh_fir_t** out);
    void gsh_fir_destroy(gsh_fir_t* f);
    int gsh_fir_process_device(gsh_fir_t* f, const void* device_in, uint64_t n_in, void* device_out, uint64_t max_out, uint64_t* n_out, void* hip_stream);
    /* bind a bank to a ring: from now on gsh_corr_job.sample_offset is an ABSOLUTE sample index; a job whose window is not
     * fully resident (or longer than max_window_samples) fails with GSH_ERR_INVALID.  NULL detaches.
     * Residency is checked when the job table is staged (gsh_bank_correlate / gsh_bank_upload_jobs): a table uploaded once and
     * launched repeatedly (gsh_bank_launch) keeps pointing at the ring positions it was translated to, so the caller must not push
     * past those windows in between.  The handles are not internally synchronised: one thread at a time per handle, and pushes
     * into a ring are serialised against the calls that read it (Hip_Correlator_Runtime does both for a receiver). */
    int gsh_bank_set_stream_ring(gsh_bank_t* b, gsh_stream_t* s);

    /* ================================================================ TRACKING LOOP (closed on the device)
     * gsh_trk_*: the steady-state loop of dll_pll_veml_tracking (trk.cc state 2, :1975-2001), one code period per
     * iteration:  do_correlation_step (trk.cc:1232-1257) -> run_dll_pll (:1260-1324: Costas / four-quadrant PLL
     * discriminator, fll_diff_atan, Tracking_FLL_PLL_filter, E-L or VEMLP DLL discriminator, Tracking_loop_filter,
     * carrier aiding) -> update_tracking_vars (:1409-1483) -> consume d_current_prn_length_samples (:2287),
     * with the initial conditions of start_tracking (:796-866) and the pull-in state (:1949-1973).
     * All of it runs on the GPU: one work-group per channel iterates over the code periods of a device-resident IF
     * stream and leaves one record per period, so n_epochs periods cost one launch instead of n_epochs host round
     * trips.  With enable_lock_detectors the C/N0 estimator, the carrier lock detector, their smoothers and the loss-of-lock
     * counters (cn0_and_tracking_lock_status, trk.cc:1167-1224; T/lock_detectors.cc, T/exponential_smoother.cc) run on the device
     * too, between the correlation and run_dll_pll as state 2 orders them (:2008-2018).  Not modelled here (the host block
     * keeps them): the experimental Doppler correction (:1326-1346).  Symbol synchronisation, narrow tracking, extended integration,
     * the histogram bit synchroniser and high dynamics are switched on by the corresponding gsh_trk_conf fields.
     * T/ = src/algorithms/tracking/libs/.
     */
    typedef struct gsh_trk gsh_trk_t;

    typedef struct gsh_trk_conf
    {
        double fs_in;                    /* Dll_Pll_Conf::fs_in */
        double code_chip_rate;           /* d_code_chip_rate [chips/s] */
        double signal_carrier_freq;      /* d_signal_carrier_freq [Hz] (carrier aiding, trk.cc:1320-1323) */
        double cfo_frequency_hz;         /* d_cfo_frequency_hz, trk.cc:1423 */
        uint32_t code_length_chips;      /* d_code_length_chips */
        uint32_t code_samples_per_chip;  /* d_code_samples_per_chip: 1, or 2 for the E1 sinBOC replica (trk.cc:289) */
        uint32_t vector_length;          /* samples per correlation, trk.cc:1243 */
        int32_t veml;                    /* 0: E/P/L, 1: VE/E/P/L/VL (trk.cc:609-650) */
        int32_t track_pilot;             /* also correlate the data-component code, 1 tap (trk.cc:1246-1256) */
        float early_late_space_chips;
        float very_early_late_space_chips;
        float pll_bw_hz, dll_bw_hz, fll_bw_hz;
        int32_t pll_filter_order;        /* 2 or 3 (T/tracking_FLL_PLL_filter.cc:23-54) */
        int32_t dll_filter_order;        /* 1..3 (T/tracking_loop_filter.cc:101-196) */
        int32_t enable_fll_pull_in, enable_fll_steady_state, carrier_aiding;
        int32_t cloop;                   /* d_cloop: 1 = Costas two-quadrant arctangent, 0 = four-quadrant */
        uint32_t pull_in_time_s;         /* Dll_Pll_Conf::pull_in_time_s, trk.cc:1912-1915 */
        float spc, slope, y_intercept;   /* dll_nc_e_minus_l_normalized parameters (trk.cc:1986, dll_pll_conf.h:55-57) */
        int32_t enable_lock_detectors;   /* 0: the loop never declares loss of lock (the host block runs the detectors) */
        int32_t cn0_samples;             /* Dll_Pll_Conf::cn0_samples, 1..GSH_MAX_CN0_SAMPLES (flag default 20) */
        int32_t cn0_min;                 /* [dB-Hz] (25) */
        int32_t max_code_lock_fail;      /* (50) */
        int32_t max_carrier_lock_fail;   /* (5000) */
        int32_t cn0_smoother_samples;    /* (200) divided by the code period in ms, trk.cc:683-686 */
        int32_t carrier_lock_test_smoother_samples; /* (25) */
        float cn0_smoother_alpha;        /* (0.002) */
        float carrier_lock_test_smoother_alpha;     /* (0.002) */
        double carrier_lock_th;          /* (0.7) */
        /* symbol synchronisation and the narrow-tracking state (trk.cc:2026-2104 and state 4, :2197-2252); 0 = the loop stays in state 2.
         * Per-signal values as the block's constructor sets them (trk.cc:196-300): GPS L1 C/A symbols_per_bit 20, secondary_code = the
         * 160-symbol telemetry preamble (GPS_CA_PREAMBLE_SYMBOLS_STR), has_secondary 0; Galileo E1 pilot symbols_per_bit 1,
         * secondary_code = the 25-chip E1C code, has_secondary 1; GPS L5 the NH codes.  extend_correlation_symbols > 1 adds the coherent
         * integration of trk.cc:2114-2149, 2156-2195: after synchronisation the loop filters are re-parameterised (dll_bw_narrow_hz and
         * pll_bw_narrow_hz over the stretched update interval, filter memories kept), the correlator spacing narrows, and the loop closes
         * once every extend_correlation_symbols periods on the accumulated correlators.  use_histogram_bit_sync adds the
         * HistogramBitSynchronizer of T/bit_synchronizer.cc in front of the preamble search, as the block configures it for signals
         * without a secondary code and more than one symbol per bit (trk.cc:1387-1406, 2046-2072). */
        int32_t enable_symbol_sync;
        int32_t symbols_per_bit;         /* d_symbols_per_bit */
        int32_t has_secondary;           /* d_secondary: in state 4 the correlators are multiplied by the secondary code chip */
        int32_t secondary_code_length;   /* d_secondary_code_length, <= GSH_MAX_SECONDARY */
        int32_t data_secondary_code_length; /* d_data_secondary_code_length */
        int32_t extend_correlation_symbols; /* Dll_Pll_Conf::extend_correlation_symbols (1) */
        uint8_t secondary_code[320];     /* characters '0' / '1' (d_secondary_code_string) */
        uint8_t data_secondary_code[320];
        float pll_bw_narrow_hz;          /* (5.0)  dll_pll_conf.h:49 */
        float dll_bw_narrow_hz;          /* (0.75) */
        float early_late_space_narrow_chips;      /* (0.15) */
        float very_early_late_space_narrow_chips; /* (0.5) */
        int32_t use_histogram_bit_sync;  /* d_use_histogram_bit_sync */
        int32_t bs_min_events_for_lock;  /* (10) dll_pll_conf.h:76 */
        int32_t bs_stable_best_required; /* (3) */
        int32_t bs_use_phase_dot_detector; /* (1) */
        float bs_min_prompt_mag;         /* (0.0) */
        int32_t enable_bit_sync_time_limit; /* 1: the fail-safe of trk.cc:2000-2007 -- a channel still in state 2 (no secondary code / bit synchronisation yet)
                                               more than bit_synchronization_time_limit_s whole seconds after the acquisition stamp is declared lost.  The
                                               adapters switch it on (the reference has no switch); needs enable_lock_detectors and enable_symbol_sync */
        double bs_dominance_ratio;       /* (0.6) */
        int32_t high_dyn;                /* Dll_Pll_Conf::high_dyn: the high-dynamics resampler + rotator (trk.cc:669-675) fed with the rate-of-change
                                            estimates of both NCO steps (trk.cc:1425-1443, 1458-1480) */
        uint32_t smoother_length;        /* (10) periods per average, <= GSH_MAX_SMOOTHER */
        uint32_t bit_synchronization_time_limit_s; /* (20) Dll_Pll_Conf::bit_synchronization_time_limit_s */
        int32_t enable_doppler_correction; /* Dll_Pll_Conf::enable_doppler_correction (false): the experimental one-shot re-initialisation of the carrier loop
                                              when the filtered code error averages more than 1 chip/s over 1000 loop updates after pull-in (trk.cc:1326-1346) */
    } gsh_trk_conf;
#define GSH_MAX_BITSYNC_BINS 64
#define GSH_MAX_SMOOTHER 32
#define GSH_MAX_CN0_SAMPLES 64
#define GSH_MAX_SECONDARY 320  /* the longest pattern the reference correlates: the 300-symbol GLONASS GNAV preamble */

    typedef struct gsh_trk_epoch         /* what log_data dumps per period (trk.cc:1599-1702), POD */
    {
        uint64_t sample_counter;         /* first sample of the correlated window */
        int32_t prn_length_samples;      /* d_current_prn_length_samples: samples consumed after this period */
        int32_t flags;                   /* bit 0: d_pull_in_transitory was set; bit 1: loss of lock declared in this period
                                            (trk.cc:1208-1221, "events" message 3): the channel stops, the record carries the
                                            correlator outputs and the detector values only, prn_length_samples = 0 */
        float corr[10];                  /* E,P,L or VE,E,P,L,VL as interleaved complex64 */
        float prompt_data[2];            /* d_Prompt_Data (track_pilot) */
        float rem_carr_phase_rad;
        float cn0_db_hz;                 /* d_CN0_SNV_dB_Hz (0 until cn0_samples periods have passed, or with the detectors off) */
        double carrier_doppler_hz, code_freq_chips, carr_phase_error_hz, carr_freq_error_hz, carr_error_filt_hz;
        double code_error_chips, code_error_filt_chips, rem_code_phase_samples, acc_carrier_phase_rad;
        double carrier_lock_test;        /* d_carrier_lock_test */
        int32_t state;                   /* d_state the period ran in: 2 wide tracking / symbol search, 3 coherent integration (no loop update:
                                            the loop fields of the record are 0), 4 narrow tracking (0 with symbol sync off) */
        int32_t symbol_flags;            /* bit 0: Flag_valid_symbol_output (a telemetry symbol leaves the block, trk.cc:2212-2236);
                                            bit 1: Flag_PLL_180_deg_phase_locked (trk.cc:1148-1156) */
        float p_data_accu[2];            /* d_P_data_accu: Prompt_I / Prompt_Q of the symbol when bit 0 is set, else the running sum */
        double carrier_phase_rate_step_rad; /* d_carrier_phase_rate_step_rad [rad/sample^2] after this period (0 outside high_dyn) */
        double code_phase_rate_step_chips;  /* d_code_phase_rate_step_chips [chips/sample^2] */
        float accu[10];                  /* d_VE_accu .. d_VL_accu as run_dll_pll / log_data saw them (E,P,L in slots 0..2 without veml): the period's
                                            outputs in state 2, the secondary-code-wiped running sums in states 3 / 4 (trk.cc:1486-1512, 1624-1636) */
    } gsh_trk_epoch;

    int gsh_trk_create(int device, const gsh_trk_conf* conf, int n_channels, int max_code_length, gsh_trk_t** out);
    void gsh_trk_destroy(gsh_trk_t* t);
    /* IF sample stream shared by every channel: _host copies, _device borrows 16-byte aligned device memory */
    int gsh_trk_set_stream_host(gsh_trk_t* t, const float* iq, uint64_t n_samples);
    int gsh_trk_set_stream_device(gsh_trk_t* t, const void* device_iq, uint64_t n_samples);
    /* follow a live sample ring: start_sample / sample_counter are absolute sample indices; every gsh_trk_run works through the
     * periods whose windows are resident at the time of the call and stops at the newest sample, the next call (after more pushes)
     * continues.  A channel whose next window has already been overwritten (it fell more than the ring's capacity behind) stops. */
    int gsh_trk_set_stream_ring(gsh_trk_t* t, gsh_stream_t* s);
    /* start_tracking (trk.cc:796-866) for one channel: local replica(s) (code_length floats = chips x samples per
     * chip; data_code NULL unless track_pilot), Acq_doppler_hz, Acq_samplestamp_samples, and the first sample of the
     * first code period, i.e. the stream position after the pull-in alignment of trk.cc:1949-1973 */
    int gsh_trk_start(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample,
        uint64_t acq_sample_stamp, double acq_carrier_doppler_hz);
    /* the same with d_acc_carrier_phase_rad preset (the pull-in alignment subtracts step * samples_offset from it, trk.cc:1966) */
    int gsh_trk_start_ex(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample,
        uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, double initial_acc_carrier_phase_rad);
    /* The pull-in arithmetic of dll_pll_veml_tracking::general_work, state 1 (trk.cc:1949-1978), host only: with the block's read pointer at
     * absolute sample nitems_read and the acquisition's Acq_delay_samples / Acq_samplestamp_samples / Acq_doppler_hz, how many samples
     * to skip so that the next sample is a code start (samples_offset -> consume_each), the nominal first block length
     * (d_current_prn_length_samples = round(T_prn_mod_samples)), and d_acc_carrier_phase_rad after the skip.
     * start_sample for gsh_trk_start_ex is nitems_read + samples_offset. */
    int gsh_trk_pull_in(const gsh_trk_conf* conf, uint64_t nitems_read, double acq_delay_samples, uint64_t acq_sample_stamp, double acq_carrier_doppler_hz,
        int32_t* samples_offset, int32_t* first_prn_length_samples, double* acc_carrier_phase_rad);
    /* The pull-in transitory (d_pull_in_transitory, trk.cc:1073, 1910-1917) is a latch that EVERY general_work call looks at -- the pull-in call (state 1) included:
     *   pull_in_time_s < (nitems_read(0) - d_acq_sample_stamp) / (int)fs_in        in unsigned 64-bit arithmetic.
     * A tracking block whose read pointer is still BEHIND the acquisition's sample stamp at that call (it runs only when two code periods of input are there, the
     * acquisition block consumes whatever it is offered: tracking behind acquisition is the normal order of things in a flowgraph) wraps that difference round to
     * ~2^64 and the transitory is over before the first period: bit synchronisation starts at once and the lock detectors count from the first test.  Returns 1 when
     * that is the case for the pull-in call at read pointer nitems_read (host only) -> GSH_TRK_START_PULL_IN_OVER for gsh_trk_start_flags. */
    int gsh_trk_pull_in_over(const gsh_trk_conf* conf, uint64_t nitems_read, uint64_t acq_sample_stamp);
#define GSH_TRK_START_PULL_IN_OVER 1u
    /* gsh_trk_start_ex + flags (GSH_TRK_START_PULL_IN_OVER: the channel starts with the pull-in transitory already over, see above) */
    int gsh_trk_start_flags(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample,
        uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, double initial_acc_carrier_phase_rad, uint32_t flags);
    /* stop_tracking / clear_tracking_vars for one channel: the loop no longer advances it (its state stays readable) */
    int gsh_trk_stop(gsh_trk_t* t, int channel);
    /* n_epochs code periods of every started channel in ONE launch; the loop state stays on the device, so a later
     * call continues where this one stopped.  records: n_channels * n_epochs (channel-major) or NULL;
     * epochs_done[n_channels]: periods completed (a channel stops when its window would leave the stream). */
    int gsh_trk_run(gsh_trk_t* t, int n_epochs, gsh_trk_epoch* records, int32_t* epochs_done);
    /* Cooperating work-groups (round 6): `work_groups_per_channel` work-groups on different compute units share every window of a channel in LAUNCHED runs
     * (gsh_trk_run / _run_begin / _time_run; standard correlator only) -- each correlates its segment of the window, the channel's main work-group adds the partial
     * sums in a fixed order and runs the loop.  For few channels on a large device; MEASURED at 32 channels, 25 Msps: 7.29 us per period with one work-group,
     * 6.75 - 7.4 with two, 7.2 - 7.8 with four, ~9 with eight -- two hand-overs through L2 per period (~0.7 us each) eat most of what the shorter
     * correlation saves, so this is an option, not the default.  The sums are
     * formed in another order than with one work-group per channel, so records agree with that form to rounding (same bars against the oracle), not bit for bit;
     * live residencies always run one work-group per channel.  Needs (ceil(n_channels / 8) x 8 x work_groups_per_channel) compute units free at once; a partner
     * that does not get to run within 0.2 s ends the run with GSH_ERR_STATE instead of hanging the device.  1 = off (default).
     * Where it pays is the LONG window: BASELINE config 4 (50 channels, 128 000 samples, 5 + 1 taps) 36.5 us per period with one work-group, 22.0 with two,
     * 16.1 with four.  0 = choose by the window's length in trips and the compute units the channels leave free (2 for 25 000-sample E/P/L windows, 4 for config 4). */
    int gsh_trk_set_split(gsh_trk_t* t, int work_groups_per_channel);
    /* the same in two halves, for a caller that serialises launches against pushes into the ring itself (Hip_Tracking_Runtime): _begin queues
     * the launch on the loop's stream -- the kernel writes its results into page-locked host memory itself (GSH_TRK_HOST_RECORDS=0 in the environment:
     * into device memory, with two copies queued behind it) -- and returns at once; it is the only part
     * that looks at the ring's bookkeeping (newest sample, reader fences), so the ring's lock is needed around it alone and the next block of
     * samples can travel while the kernel runs; _end waits for the stream and hands the results over.  One launch in flight per handle. */
    int gsh_trk_run_begin(gsh_trk_t* t, int n_epochs, int want_records);
    int gsh_trk_run_end(gsh_trk_t* t, gsh_trk_epoch* records, int32_t* epochs_done);
    /* ---- live mode: the loop follows the ring as it fills, without a launch per batch of periods.
     * In the reference every channel's block is called once per code period (trk.cc:1898-2001) off one shared buffer
     * (src/core/receiver/gnss_flowgraph.cc:1227-1231); at that cadence a launch per call is all host time.  gsh_trk_live_begin queues a RESIDENCY of the loop
     * kernel instead: every started channel correlates its next window as soon as the ring (gsh_trk_set_stream_ring) holds it -- pushes announce
     * themselves to the resident kernel through the ring, no event, no host --, and leaves one record per period in a ring of records in page-locked
     * host memory that gsh_trk_live_take reads without a lock or a device call.  A residency ends by itself when nothing new has arrived for
     * idle_timeout_us (1000), after residency_us (20000: anything that waits for the whole device gets its turn), when the host asks (gsh_trk_live_quiesce)
     * or when no channel has work; up to two may be queued, the second takes over when the first ends.  Loop arithmetic, records and trajectories are
     * those of gsh_trk_run, bit for bit.
     * Threads: _begin, _in_flight, _quiesce, start, stop, run*: one at a time per handle (the caller's lock).  _take: any thread, at most one per
     * CHANNEL at a time, concurrently with everything except start / stop of that channel and gsh_trk_destroy.
     * Pushes into the ring never overwrite samples a live channel has not correlated yet: they fail with GSH_ERR_STATE instead (an event cannot order
     * a push behind a kernel that is waiting for that push); the caller keeps its pushes below lowest next window + ring capacity. */
    int gsh_trk_live_configure(gsh_trk_t* t, uint32_t idle_timeout_us, uint32_t residency_us);
    int gsh_trk_live_begin(gsh_trk_t* t);
    int gsh_trk_live_in_flight(gsh_trk_t* t, int32_t* residencies);  /* queued or running */
    /* up to max_records finished periods of one channel, oldest first, whose samples lie below limit_end (sample_counter + max(vector_length,
     * prn_length_samples) <= limit_end; UINT64_MAX: no limit).  A record with flags bit 1 (loss of lock) ends the take and is the channel's last.
     * *pending: records finished and not taken; *next_window: first sample of the window behind the last record taken; *active: the device still
     * advances the channel; *resident: the channel's work-group is inside a running residency right now (0: it has left -- idle, time budget, quit --
     * or none has started yet: a caller that waits for a record then knows to ask for one, without a device call). */
    int gsh_trk_live_take(gsh_trk_t* t, int channel, uint64_t limit_end, int max_records, gsh_trk_epoch* out, int32_t* n_out, int32_t* pending,
        uint64_t* next_window, int32_t* active, int32_t* resident);
    /* tell the residencies in flight to leave, wait for them; records not yet taken stay where they are.  Needed before start / stop / run. */
    int gsh_trk_live_quiesce(gsh_trk_t* t);
    /* where every channel stands after the last completed run (host copy, refreshed by gsh_trk_run / _run_end and by start / stop):
     * next_window[ch] = absolute index of the first sample of the channel's next correlation window, active[ch] != 0 while the loop advances it */
    int gsh_trk_positions(gsh_trk_t* t, uint64_t* next_window, int32_t* active);
    /* HIP-event milliseconds of one gsh_trk_run-sized launch, averaged over reps (each rep restarts from the state at
     * entry; the state is restored afterwards) */
    int gsh_trk_time_run(gsh_trk_t* t, int n_epochs, int reps, float* avg_ms);
    /* write (append != 0: append) the records of one channel as a tracking dump file in the block's own binary layout
     * (log_data, trk.cc:1599-1702: 19 floats, the PRN start sample as uint64 and as double, PRN, TOW [ms] as uint64, week number =
     * 108 bytes per logged period -- the layout save_matfile (trk.cc:1705-1716) and utils/matlab/libs/dll_pll_veml_read_tracking_dump.m
     * parse).  Host-only.  Logged are the periods the block logs: every period of state 2, the symbol periods of states 3 / 4
     * (trk.cc:2024, 2166, 2218); periods flagged as loss of lock are skipped, as the reference skips log_data there.  tow_ms / wn: one
     * value per record (d_tow_from_telemetry_ms / d_wn_from_telemetry of that call, trk.cc:1921-1935) or NULL for zeros. */
    int gsh_trk_write_dump(const char* path, int append, const gsh_trk_conf* conf, uint32_t prn, const gsh_trk_epoch* records, int n_records,
        const uint64_t* tow_ms, const uint32_t* wn);

    /* ================================================================ ACQUISITION
     * gsh_acq_*: the arithmetic of class pcps_acquisition (acq.h:93-251) without its
     * GNU Radio shell: set_local_code (acq.cc:218-251), update_grid_doppler_wipeoffs
     * (acq.cc:284-291), doppler_grid (acq.cc:522-560), both statistics (acq.cc:409-519).
     * One handle searches n_prn local codes against the same input block in one go
     * (the reference runs one block per channel, each recomputing the D forward FFTs).
     */
    typedef struct gsh_acq gsh_acq_t;

    typedef struct gsh_acq_conf
    {
        int64_t fs_in;              /* Acq_Conf::fs_in (resampled_fs when the resampler is on), acq.cc:277 */
        uint32_t fft_size;          /* d_fft_size, acq.cc:111.  Any length: on-chip plans for the common ones, a four-step split for prime factors
                                       <= 61, and for the rest (plain searches) a zero-padded power-of-two form of the same circular correlation */
        uint32_t effective_fft_size; /* d_effective_fft_size, acq.cc:112 */
        uint32_t consumed_samples;  /* d_consumed_samples, acq.cc:110 */
        uint32_t num_doppler_bins;  /* d_num_doppler_bins, acq.cc:113; 0 = ceil(2*doppler_max/doppler_step) */
        int32_t doppler_max;        /* Acq_Conf::doppler_max */
        int32_t doppler_step;       /* Acq_Conf::doppler_step */
        int32_t doppler_center;     /* set_doppler_center, acq.cc:737-746 */
        int32_t doppler_bias;       /* GLONASS FDMA offset, acq.cc:254-272; 0 otherwise */
        uint32_t samples_per_chip;  /* Acq_Conf::samples_per_chip, acq_conf.cc:122 */
        float samples_per_code;     /* Acq_Conf::samples_per_code, acq_conf.cc:123 */
        int32_t bit_transition_flag; /* acq.cc:230-235,544 */
        int32_t use_cfar;           /* d_use_CFAR_algorithm_flag: 1 = max_to_input_power, 0 = first_vs_second_peak */
        uint32_t max_prn;           /* how many local codes this handle can hold */
        int32_t no_grid;            /* 0: the |.|^2 grid d_magnitude_grid (acq.cc:137) is kept in device memory, as the
                                       reference keeps it (needed by accumulate != 0, i.e. max_dwells > 1, and by
                                       gsh_acq_read_grid, i.e. dump).  1: the caller promises neither; lengths with an
                                       on-chip plan then never write the grid (statistics are formed on chip) */
        int32_t transform_path;     /* 0: automatic (whole transform on one CU when the length has a plan, else the
                                       four-step path through HBM); 1: force the four-step path (A/B testing) */
        uint32_t num_doppler_bins_step2; /* Acq_Conf::num_doppler_bins_step2 (acq_conf.h:62, key second_nbins); 0 = the handle
                                       never runs the fine-Doppler step (make_two_steps = false); must be <= num_doppler_bins */
        float doppler_step2;        /* Acq_Conf::doppler_step2 (acq_conf.h:50, key second_doppler_step) */
        uint32_t fold;              /* 0 or 1: off.  > 1 (QuickSync, pcps_quicksync_acquisition_cc.cc:243-263): the input block is
                                       fold * fft_size samples (= consumed_samples); after the Doppler wipe-off its `fold` segments
                                       are added into one fft_size-long block before the forward transform.  The block passes
                                       folding_factor^2; set_local_code takes the code already folded to fft_size (:137-152) */
    } gsh_acq_conf;

    typedef struct gsh_acq_result
    {
        uint32_t index_time;        /* AcquisitionResult::index_time  (lowest index on ties) */
        uint32_t index_doppler;     /* winning Doppler bin (first bin on ties, acq.cc:420) */
        int32_t doppler_hz;         /* AcquisitionResult::doppler, acq.cc:431 */
        float acq_delay_samples;    /* fmod((float)index_time, samples_per_code), acq.cc:582 */
        float peak;                 /* grid maximum */
        float input_power;          /* d_input_power (CFAR only), acq.cc:430 */
        float second_peak;          /* peak-ratio statistic only, acq.cc:511-513 */
        float test_statistics;      /* acq.cc:439-445 or :516 */
    } gsh_acq_result;

    int gsh_acq_create(int device, const gsh_acq_conf* conf, gsh_acq_t** out);
    void gsh_acq_destroy(gsh_acq_t* a);
    /* acq.cc:218-251: place the time-domain replica per the padding rules, FFT, conjugate.
     * `code_iq` holds consumed_samples complex samples (fft_size/2 when bit_transition_flag). */
    int gsh_acq_set_local_code(gsh_acq_t* a, uint32_t prn_slot, const float* code_iq);
    /* acq.cc:737-746 */
    int gsh_acq_set_doppler_center(gsh_acq_t* a, int32_t doppler_center);
    /* acq.cc:252-272 (is_fdma): GLONASS L1 / L2 satellites sit on their own FDMA carrier, DFRQ{1,2}_GLO * channel number away from
     * the band centre; the offset is re-derived whenever the PRN changes and enters only the wipe-off frequency (acq.cc:289), never
     * the reported Doppler.  Applies to every prn slot of the handle: search GLONASS satellites one frequency channel per handle. */
    int gsh_acq_set_doppler_bias(gsh_acq_t* a, int32_t doppler_bias);
    /* The other PCPS detectors on the same engine:
     * pcps_tong_acquisition_cc.cc:243-249 scales every |y|^2 by 1 / (fft_norm^2 * input_power) BEFORE adding it to its
     * d_grid_data; `weight` is that factor, applied (one float multiply per cell) to every magnitude that is added to or
     * stored in the grid by the dwells that follow.  Default 1 (the multiply is then exact).  Needs no_grid == 0 when != 1. */
    int gsh_acq_set_grid_weight(gsh_acq_t* a, float weight);
    /* mean |x|^2 of the consumed_samples block most recently handed to the handle -- d_input_power of
     * pcps_tong_acquisition_cc.cc:208-210 and galileo_pcps_8ms_acquisition_cc.cc:190-192.  Per-sample terms are the
     * reference's float values; they are summed in double (the reference's float VOLK accumulator has no defined lane
     * order), so the value agrees with the reference to float rounding of the sum (~1e-7 relative), not bit for bit. */
    int gsh_acq_input_power(gsh_acq_t* a, float* mean_power);
    /* The Tong weight of a block depends on that block's own power, so the block must be resident before its dwell is
     * queued: stage_input[_device] copies consumed_samples complex64 samples into the handle (what gsh_acq_dwell[_device]
     * do first), dwell_resident then runs the dwell of gsh_acq_dwell over the resident block. */
    int gsh_acq_stage_input(gsh_acq_t* a, const float* in_iq);
    int gsh_acq_stage_input_device(gsh_acq_t* a, const void* device_in_iq);
    int gsh_acq_dwell_resident(gsh_acq_t* a, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results);
    /* per-Doppler-bin maximum and its (lowest) time index of prn_slot's grid after the last full dwell -- magt / indext of
     * the per-bin loops of galileo_pcps_8ms_acquisition_cc.cc:226-262, which compares two local codes bin by bin.
     * num_doppler_bins entries each. */
    int gsh_acq_read_row_peaks(gsh_acq_t* a, uint32_t prn_slot, float* row_peak, uint32_t* row_index_time);
    /* Non-coherent I + Q combination of galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc:357-492 after a dwell over the block's local
     * codes held in four slots of this handle: data component "A" (1,1,1) in slot_ia, pilot "A" in slot_qa (-1: data only,
     * d_both_signal_components false), and for coherent times above one code period the "B" combinations with the first period
     * inverted in slot_ib / slot_qb (-1: d_sampled_ms == 1).  Per Doppler bin the component's A or B row is kept as the block keeps it
     * (row maxima / N^4 compared with >=, :403, :410 -- the Q-B candidate ranked by the I-B row at Q-B's arg-max, as :393 is written),
     * the two kept magnitude rows are ADDED cell by cell on the device (:419-431) and the maximum of the sum with its lowest index is
     * returned (:487-492), together with the two row maxima the CAF filter works on (:405-427).  num_doppler_bins records.
     * Needs the stored grid (no_grid = 0). */
    typedef struct gsh_acq_pair_peak
    {
        float peak;           /* d_magnitudeI[indext] after the addition, unnormalised */
        uint32_t index_time;  /* indext */
        float caf_i;          /* d_CAF_vector_I[doppler_index] */
        float caf_q;          /* d_CAF_vector_Q[doppler_index] (0 without a pilot slot) */
        uint32_t i_slot;      /* the slots whose rows were added */
        uint32_t q_slot;      /* 0xFFFFFFFF: none */
    } gsh_acq_pair_peak;
    int gsh_acq_noncoherent_pair_peaks(gsh_acq_t* a, int32_t slot_ia, int32_t slot_qa, int32_t slot_ib, int32_t slot_qb, gsh_acq_pair_peak* out);
    /* QuickSync's de-ambiguation (pcps_quicksync_acquisition_cc.cc:295-323): time-domain correlation of the resident block,
     * wiped off at Doppler bin `doppler_index`, with the UNfolded code (code_len complex64 samples, host memory) at n_delays
     * (<= 100) candidate delays: out[c] = sum_j x[delays[c] + j] w[delays[c] + j] code[j].  Float products as the reference forms
     * them, summed in double (the reference adds sequentially in float: agreement ~1e-6 relative, not bit for bit). */
    int gsh_acq_time_correlate(gsh_acq_t* a, const float* code_iq, uint32_t code_len, uint32_t doppler_index, const uint32_t* delays,
        uint32_t n_delays, float* out_iq);
    /* one acquisition_core pass (acq.cc:648-684) for prn_slot 0..n_prn-1 over the same
     * consumed_samples input block.  `accumulate` != 0 adds to the stored grids
     * (non-coherent dwell number > 1, acq.cc:549-553); `dwell_count` is
     * d_num_noncoherent_integrations_counter after the increment (acq.cc:668), used by the
     * CFAR power normalisation (acq.cc:430). */
    int gsh_acq_dwell(gsh_acq_t* a, const float* in_iq, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results);
    /* One dwell for an arbitrary subset of the handle's local codes: results[i] belongs to prn_slots[i].  First dwell of a search only
     * (accumulate = 0, dwell count 1) and statistics only: whatever grid the handle stores is indexed by position i for this call, not by slot.
     * This is what lets the channels of a receiver that search the same input block at the same time share one batch -- the D forward transforms
     * are computed once for all of them instead of once per channel (Hip_Acquisition_Runtime; the reference runs one pcps_acquisition block per
     * channel, each recomputing them, acq.cc:522-560). */
    int gsh_acq_dwell_slots(gsh_acq_t* a, const float* in_iq, uint32_t n, const uint32_t* prn_slots, gsh_acq_result* results);
    /* same with the input block already in device memory (16-byte aligned) */
    int gsh_acq_dwell_device(gsh_acq_t* a, const void* device_in_iq, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results);
    /* Step two of make_two_steps (acq.cc:294-301, 522-560 with d_step_two, 428-437 / 475-482): for each i < n a narrow
     * grid of num_doppler_bins_step2 bins at  center[i] + ((float)d - floor(nbins2/2)) * doppler_step2  (float arithmetic,
     * acq.cc:298-299) for local code prn_slots[i] over one consumed_samples block.  `doppler_center_step_two[i]` is the
     * step-one result's Doppler (acq.cc:619); `input_power_step_one[i]` is d_input_power left by step one -- the CFAR
     * statistic of step two divides by it without recomputing it (acq.cc:428-445); ignored for the peak-ratio statistic.
     * results[i].doppler_hz = (int32_t)(center + ((float)index_doppler - floor(nbins2/2)) * doppler_step2) (acq.cc:436).
     * `accumulate`/`dwell_count` as in gsh_acq_dwell (non-coherent dwells inside step two, acq.cc:545-553). */
    int gsh_acq_dwell_step2(gsh_acq_t* a, const float* in_iq, uint32_t n, const uint32_t* prn_slots, const float* doppler_center_step_two,
        const float* input_power_step_one, int accumulate, uint32_t dwell_count, gsh_acq_result* results);
    int gsh_acq_dwell_step2_device(gsh_acq_t* a, const void* device_in_iq, uint32_t n, const uint32_t* prn_slots,
        const float* doppler_center_step_two, const float* input_power_step_one, int accumulate, uint32_t dwell_count, gsh_acq_result* results);
    /* item_type = cshort (acq.cc:653-656): `in_iq16` holds consumed_samples interleaved int16 I,Q pairs, converted to
     * complex64 on the device (volk_gnsssdr_16ic_convert_32fc: exact int -> float), then as gsh_acq_dwell */
    int gsh_acq_dwell_cshort(gsh_acq_t* a, const int16_t* in_iq16, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results);
    /* the input block is consumed_samples resident samples of a device ring starting at absolute index first_sample (the ring's
     * max_window_samples must cover consumed_samples): what the acquisition block copies out of its input buffer (acq.cc:790-815) */
    int gsh_acq_dwell_ring(gsh_acq_t* a, gsh_stream_t* ring, uint64_t first_sample, uint32_t n_prn, int accumulate, uint32_t dwell_count,
        gsh_acq_result* results);
    /* dump support (acq.cc:555-558): copy |.|^2 grid of one PRN, D rows of effective_fft_size floats */
    int gsh_acq_read_grid(gsh_acq_t* a, uint32_t prn_slot, float* grid);
    /* HIP-event average milliseconds per full dwell batch (n_prn codes), inputs resident */
    int gsh_acq_time_dwells(gsh_acq_t* a, uint32_t n_prn, int reps, float* avg_ms);
    /* the same stream of `reps` dwell batches issued alternately on two HIP streams with per-batch spectra / row /
     * result buffers, so batch k+1's forward transforms and first cells run on the compute units batch k's tail
     * leaves idle: average milliseconds per batch in steady state (throughput; the single-batch latency is what
     * gsh_acq_time_dwells reports).  Statistics only (as no_grid = 1); lengths without an on-chip plan fall back to
     * gsh_acq_time_dwells. */
    int gsh_acq_time_dwells_pipelined(gsh_acq_t* a, uint32_t n_prn, int reps, float* avg_ms);

    /* ---- pulse blanking (input filter in front of the channels) ---------------------------------------------------------------
     * pulse_blanking_cc (src/algorithms/input_filter/gnuradio_blocks/pulse_blanking_cc.cc:33-106; adapter key
     * InputFilter.implementation=Pulse_Blanking_Filter, pulse_blanking_filter.cc: pfa, length, segments_est, segments_reset): the
     * stream is tiled into `length`-sample segments; a segment whose energy over the estimated noise power exceeds the chi-squared
     * (2 * length degrees of freedom) threshold for `pfa` is zeroed.  State (noise estimate, segment counter, last_filtered) lives on
     * the device and carries over between calls. */
    typedef struct gsh_pulse_blanking gsh_pb_t;
    int gsh_pb_create(int device, float pfa, int32_t length, int32_t n_segments_est, int32_t n_segments_reset, gsh_pb_t** out);
    void gsh_pb_destroy(gsh_pb_t* p);
    float gsh_pb_threshold(const gsh_pb_t* p); /* thres_, pulse_blanking_cc.cc:48-49 */
    /* one general_work call (:57-106) over n_items resident complex64 samples: whole segments are filtered while
     * (index + length) < n_items, exactly as the block consumes them; *n_done = samples consumed = samples produced (the caller
     * presents the remainder again, followed by new samples, like the GNU Radio scheduler does).  In-place (out == in) is allowed. */
    int gsh_pb_process_device(gsh_pb_t* p, const void* device_in_iq, uint64_t n_items, void* device_out_iq, uint64_t* n_done);
    int gsh_pb_get_state(gsh_pb_t* p, float* noise_power_estimation, int32_t* n_segments, int32_t* last_filtered);

    /* Notch (src/algorithms/input_filter/gnuradio_blocks/notch_cc.cc:33-140; adapter Notch_Filter: pfa, p_c_factor, length, segments_est,
     * segments_reset) and NotchLite (.../notch_lite_cc.cc:30-150; adapter Notch_Filter_Lite, n_segments_coeff = coeff_rate-derived) -- continuous-wave
     * interference excision: segments of `length` samples whose energy over the estimated noise floor exceeds the chi-squared threshold for `pfa`
     * go through  out[n] = in[n] - z0 in[n-1] + p_c_factor z0 out[n-1];  the others pass unchanged.  n_segments_coeff = 0: Notch (z0 per sample
     * from the phase of in[n] conj(in[n-1])); >= 1: NotchLite (one z0 per that many filtered segments).  The floor estimate (FFT-based spectral
     * noise floor of the first n_segments_est segments after each reset), the segment counter, the filter state and the last output live on the
     * device and carry over between calls. */
    typedef struct gsh_notch gsh_notch_t;
    int gsh_notch_create(int device, float pfa, float p_c_factor, int32_t length, int32_t n_segments_est, int32_t n_segments_reset, int32_t n_segments_coeff,
        gsh_notch_t** out);
    void gsh_notch_destroy(gsh_notch_t* p);
    float gsh_notch_threshold(const gsh_notch_t* p); /* thres_, notch_cc.cc:54-55 */
    /* one general_work call over n_items resident complex64 items, of which item 0 is the sample IN FRONT of the first one processed (notch_cc.cc:71
     * `in++`; NotchLite's set_history(2) item): whole segments are taken while (index + length) < n_items; *n_done = items consumed = samples produced
     * (out[k] is the filtered in[k + 1]); the caller presents the rest again like the GNU Radio scheduler does.  Not in place. */
    int gsh_notch_process_device(gsh_notch_t* p, const void* device_in_iq, uint64_t n_items, void* device_out_iq, uint64_t* n_done);
    int gsh_notch_get_state(gsh_notch_t* p, float* noise_pow_est, int32_t* n_segments, int32_t* filter_state, float* last_out_iq, int32_t* n_segments_coeff,
        float* z0_iq);

    /* Fine-Doppler step of pcps_acquisition_fine_doppler_cc (gnuradio_blocks/pcps_acquisition_fine_doppler_cc.cc:316-389): the
     * n complex64 samples x (host), multiplied element-wise by w when w != NULL (the aligned code replica: code wipe-off, :348), are
     * zero-padded to fft_size, transformed, and the index of the largest |X[k]|^2 (lowest k among equal maxima, :354-358) is returned;
     * `peak` (nullable) receives that |X|^2.  fft_size needs a four-step split (n1 <= 1024, n2 <= 2048, prime factors <= 61):
     * up to ~2 M points, i.e. the block's 80 x samples_per_ms up to 25 Msps. */
    int gsh_spectrum_peak(int device, const float* x_iq, const float* w_iq, uint32_t n, uint32_t fft_size, uint32_t* index, float* peak);

    /* compute_threshold, acq.cc:52-56: 2*gamma_p_inv(2*max_dwells, (1-pfa)^(1/(effective*bins))) */
    float gsh_acq_compute_threshold(float pfa, uint32_t effective_fft_size, uint32_t num_doppler_bins, uint32_t max_dwells);

#ifdef __cplusplus
}
#endif
#endif /* GNSS_SDR_HIP_H */
