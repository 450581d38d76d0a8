/*
 * sift_hip.h — C ABI of the MI355X-native SIFT hot path (libsift_hip.so).
 *
 * This is the drop-in boundary for snowiow/SIFT's `sift::Sift::calculate()`
 * (/root/reference/sift.hpp:78, sift.cpp:19-57; the only caller is main.cpp:56-57).
 * Everything the reference does between receiving the greyscale float image and returning the
 * `std::vector<InterestPoint>` — Gaussian pyramid, DoG, scale-space extrema, edge-response
 * filter, gradient maps, orientation assignment, descriptors (sift.cpp + algorithms.cpp, which
 * the reference runs through Vigra on one CPU thread) — runs here as hand-written HIP kernels
 * for gfx950.  Plain pointers and sizes only; no C++ or torch types cross this boundary.  The
 * C++ `sift::Sift` class in include/sift/sift.hpp and the Python mirror in sift_amd/ are thin
 * hosts above these entry points.
 *
 * Image layout everywhere: float32, x fastest, img(x, y) = data[x + y*w] — the layout of the
 * reference's vigra::MultiArray<2, float>.  Batches are n such frames back to back.
 *
 * Error model (reference: C++ exceptions out of calculate(), sift.cpp / Vigra preconditions):
 * every call returns a status; SIFT_HIP_EPRECONDITION carries the text of the
 * vigra::PreconditionViolation the reference would have thrown (e.g.
 * "separableConvolveY(): kernel longer than line") in `err`; SIFT_HIP_EASSERT stands for the
 * reference's assert()s (sift.cpp:382-383).
 */
#ifndef SIFT_HIP_H
#define SIFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT_HIP_OK 0
#define SIFT_HIP_EPRECONDITION 1 /* vigra::PreconditionViolation analogue, message in err */
#define SIFT_HIP_EASSERT 2       /* reference assert(): octaves == 0 or dogsPerEpoch < 3 */
#define SIFT_HIP_EINVAL 3        /* bad argument to this API */
#define SIFT_HIP_EHIP 4          /* HIP runtime failure, message in err */

typedef struct sift_hip_ctx sift_hip_ctx;

/* Constructor arguments of sift::Sift (sift.hpp:66-71), same meaning and order of magnitude
 * defaults: dogsPerEpoch 3, octaves 3, sigma 1.6, k sqrt(2), subpixel false. */
typedef struct sift_hip_params {
    uint16_t dogs_per_epoch;
    uint16_t octaves;
    float sigma;
    float k;
    uint8_t subpixel;
    uint8_t reserved[3];
} sift_hip_params;

/* POD image of sift::InterestPoint (interestpoint.hpp:13-63) without the descriptor vector;
 * descriptors travel as a separate n x 128 float array (all zero and has_descriptor == 0 for a
 * point the descriptor stage filtered, sift.cpp:65-70). */
typedef struct sift_hip_keypoint {
    float scale;
    float orientation;
    uint16_t x, y;      /* loc, octave coordinates */
    uint16_t octave;
    uint16_t index;     /* DoG index inside the octave */
    uint8_t filtered;
    uint8_t has_descriptor;
    uint16_t reserved;
} sift_hip_keypoint;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* One context per GPU / worker thread (a Sift instance is not re-entrant either, sift.hpp:46-56). */
int sift_hip_create(int device, sift_hip_ctx** out, char* err, int errlen);
void sift_hip_destroy(sift_hip_ctx* ctx);
/* Options.  The shipped library knows EIGHT names (any other returns SIFT_HIP_EINVAL); values belong to the context they are
 * set on, nothing is process-wide, and the library reads no environment variable:
 *   "profile"          N > 0: the blur launches of every N-th batch carry timing events on their dispatch packets, read with
 *                      sift_hip_profile_get (the events keep consecutive launches ~10 us apart, so a measurement run samples,
 *                      e.g. N = 4); 0 (default): off
 *   "wire_count"       1: the descriptor kernel also counts the floats the sparse wire format will carry, so that
 *                      sift_hip_result_sparse_size needs no pass of its own (hosts that gather every batch); 0 default
 *   "host_threads"     threads of the host-side copies and of the std::sort fallback (0 default: the hardware's)
 *   "spin_wait"        1 default: the end of a batch is awaited by polling its event (tens of microseconds sooner than
 *                      sleeping in hipStreamSynchronize, which 0 selects)
 *   "orient_general"   1: orientationHistogram36 reads every sample's bin even when the gradient pass found all bins of the
 *                      frame to be 0 - which is what the reference's radians-as-degrees maps always give (0 default)
 *   "stream_min_waves" smallest launch, in waves, that takes the streaming blur instead of the LDS-tiled one (0 = the default,
 *                      1024; the parity tests set 1 to run the streaming kernels on small inputs)
 *   "blur_pair"        1 default: g(0,0) and g(0,1) - two blurs with the same taps, the second of the first's result - are ONE
 *                      launch whose second stage takes the first stage's rows from LDS (sift_amd/csrc/kernels_pair.hip) when
 *                      the batch fills the chip; 0: two launches
 *   "pair_waves"       waves that launch is cut into (0 = the default, 1536)
 * Everything else the library decides by itself from the shape: rows that are not 16-byte aligned get the unfused extremum
 * scan over DoG levels the pyramid writes, radii beyond 32 the two-pass blur, index maps without a parity split the streaming
 * decimating blur, an introsort that hits its depth limit libstdc++'s std::sort on the host.  The names that FORCE those paths
 * ("fused_blur", "fused_edge", "fused_reduce", "reduce_kept", "dog_in_extrema", "gpu_cleanup", "pyramid_side",
 * "gate_schedule") and the host-side measurement aids ("diag_repeat", "diag_pyramid_span", "diag_serial_gradient",
 * "stream_waves") exist only in libsift_hip_diag.so - the shipped kernels with context.cpp compiled -DSIFT_HIP_DIAG, built by
 * `make -C sift_amd/csrc`, used by tests/diag_fallbacks.py and tools/ -, the switches inside kernels ("desc_dbg", "orient_dbg",
 * "diag_cleanup_stamps": timing only, WRONG results) only in `make ablate`'s libsift_hip_ablate.so.  Removed in round 6 with
 * their kernels (measured, not kept: DESIGN.md section 7): "tail_async", "tail_kernel", "desc_kernel", "lazy_top". */
int sift_hip_set_option(sift_hip_ctx* ctx, const char* name, int value);

/* ---- several batches in flight on one GPU --------------------------------------------------------
 * The reference runs one calculate() at a time (main.cpp:56-57).  A host that has the next batch ready can give
 * every batch in flight a context of its own (one host thread each) and join the contexts with a gate: the gate
 * orders the phases of consecutive batches on the device: this batch's cleanup steps (one workgroup per image,
 * sift.cpp:37-54), which cannot fill the chip, run under the next batch's pyramid and its descriptors under the next
 * extrema / gradient pass; no two pyramids and no two descriptor stages share the chip (sift_amd/csrc/phase_gate.h).  Results are unchanged.  Batches take their
 * place in the order in which their calculate calls begin. */
typedef struct sift_hip_gate sift_hip_gate;
int sift_hip_gate_create(int device, sift_hip_gate** out);
/* The host's handle goes at once; the gate itself lives until the last context joined by it has left (sift_hip_set_gate(ctx,
 * NULL) or sift_hip_destroy), so contexts and gate may be destroyed in either order.  No context can join a gate after this. */
void sift_hip_gate_destroy(sift_hip_gate* gate);
/* NULL detaches; not while a batch is running.  SIFT_HIP_EINVAL for a gate of another device and for the fifth
 * context on one gate (a gate tells at most four batches in flight apart). */
int sift_hip_set_gate(sift_hip_ctx* ctx, sift_hip_gate* gate);
/* Contexts that run side by side need hardware queues of their own: the HOST exports GPU_MAX_HW_QUEUES=8 (the HIP runtime's
 * default is 4) before its first HIP call - the library never writes the environment (setenv beside a host's other threads
 * is not safe).  sift_amd/_lib.py does it at import, the example programs at the top of main(). */

/* ---- host memory the copy engines can reach directly ------------------------------------------------------
 * calculate / result_copy accept any host pointer.  Memory from sift_hip_host_alloc (page-locked) moves at the PCIe rate
 * in one asynchronous copy; ordinary (pageable) memory is moved in chunks through the context's own pinned staging
 * buffers with the host-side copies overlapped, about half as fast.  NULL when the allocation fails. */
void* sift_hip_host_alloc(size_t bytes);
void sift_hip_host_free(void* p);

/* ---- Sift::calculate(), replaces sift.cpp:19-57 ---------------------------------------------- */
/* n frames of w x h from HOST memory.  Results stay in the context until the next calculate. */
int sift_hip_calculate_batch(sift_hip_ctx* ctx, const float* host_imgs, int n, int w, int h,
                             const sift_hip_params* params, char* err, int errlen);
/* Same, frames already resident in DEVICE memory of ctx's GPU (not modified). */
int sift_hip_calculate_batch_device(sift_hip_ctx* ctx, const void* dev_imgs, int n, int w, int h,
                                    const sift_hip_params* params, char* err, int errlen);

/* 8-bit frames (the reference's inputs are 8-bit files, main.cpp:52-54 vigra::importImage): a quarter of the bytes cross the
 * link and the GPU widens them to the same integer-valued floats importImage yields, so the results are those of the float
 * entry points on (float)pixel, bit for bit.  _u8 takes HOST memory, _device_u8 memory of ctx's GPU. */
int sift_hip_calculate_batch_u8(sift_hip_ctx* ctx, const uint8_t* host_imgs, int n, int w, int h,
                                const sift_hip_params* params, char* err, int errlen);
int sift_hip_calculate_batch_device_u8(sift_hip_ctx* ctx, const void* dev_imgs, int n, int w, int h,
                                       const sift_hip_params* params, char* err, int errlen);

/* Images of the last batch whose results the context holds; -1 when it holds none (no batch yet, or the last calculate
 * call failed before it ran: a failed call never leaves an earlier batch's results readable). */
int sift_hip_result_images(sift_hip_ctx* ctx);
/* Per-image status of the last batch (an image that "threw" has count 0).  `cap` = entries the caller's array holds;
 * SIFT_HIP_EINVAL if that is fewer than sift_hip_result_images(). */
int sift_hip_result_status(sift_hip_ctx* ctx, int32_t* status, int cap);
/* Number of returned InterestPoints per image, and their sum. */
int sift_hip_result_counts(sift_hip_ctx* ctx, int32_t* counts, int cap);
int64_t sift_hip_result_total(sift_hip_ctx* ctx);
/* Copy results (keypoints concatenated in image order, descriptors 128 floats each) to caller
 * buffers, which may be host or device memory. */
int sift_hip_result_copy(sift_hip_ctx* ctx, sift_hip_keypoint* keypoints, float* descriptors);
/* Device-resident packed results (valid until the next calculate) for a GPU-side gather. */
int sift_hip_result_device(sift_hip_ctx* ctx, const void** dev_keypoints, const void** dev_descriptors);
/* Wire format of a multi-GPU gather (lossless): per keypoint a 34-byte record = the 20-byte sift_hip_keypoint + 112
 * presence bits (bit cell*7+bin <-> descriptor float cell*8+bin; bin 7 is never set, algorithms.cpp:135-150), and
 * only the descriptor floats whose bit pattern is not +0.0f, in order (about a third of them on real frames).
 * _size runs the counting pass and returns the number of floats; _pack then writes total*34 bytes and that many
 * floats to DEVICE memory of the caller.  Both return when the device is done.
 * The format drops bin 7 of every cell, which is +0.0f unless a cell's bin sum is negative or NaN (normalizeVector then
 * turns the never-written bin into -0.0f or NaN: frames with negative pixels, inf * 0).  *lossless (may be NULL) is 0 when
 * the current results hold such a value: send the plain arrays of sift_hip_result_device instead. */
int sift_hip_result_sparse_size(sift_hip_ctx* ctx, int64_t* n_values, int* lossless);
int sift_hip_result_sparse_pack(sift_hip_ctx* ctx, void* dev_records, void* dev_values);
/* _pack without the wait, for a host that gathers every batch of a pipeline: the pack is queued on the context's side stream
 * and the call returns; the context's NEXT batch may be started at once (its descriptor stage - the first thing that rewrites
 * the arrays the pack reads - waits for the pack on the device, never the host).  sift_hip_result_pack_wait returns once the
 * lists are complete in dev_records / dev_values; it touches nothing but that event and may be called from another thread
 * while the context's own thread is inside its next sift_hip_calculate_batch*.  With option wire_count the batch itself
 * counts and scans (sift_hip_result_sparse_size then returns without a pass or a wait). */
int sift_hip_result_sparse_pack_async(sift_hip_ctx* ctx, void* dev_records, void* dev_values);
int sift_hip_result_pack_wait(sift_hip_ctx* ctx);
/* The same lists to HOST memory (page-locked memory moves at the link's rate): packed on the GPU, so ~200 instead of 532 bytes
 * per keypoint cross the link.  After sift_hip_result_sparse_size (use sift_hip_result_copy when it reports lossless = 0);
 * records: sift_hip_result_total() * 34 bytes, values: n_values floats. */
int sift_hip_result_copy_sparse(sift_hip_ctx* ctx, void* records, float* values);
/* ... and back, in host memory, no GPU: n_keypoints records + their floats -> sift_hip_keypoint records and n_keypoints * 128
 * descriptor floats (either may be NULL), bit for bit what sift_hip_result_copy delivers.  `threads` host threads share the
 * work (<= 1: the calling thread). */
int sift_hip_sparse_unpack_host(const void* records, const float* values, int64_t n_keypoints, sift_hip_keypoint* keypoints,
                                float* descriptors, int threads);
/* The receiving side, on ctx's GPU: n_keypoints records of 34 bytes + their floats (both in DEVICE memory, as _pack wrote them
 * on any GPU) -> n_keypoints sift_hip_keypoint records and n_keypoints * 128 descriptor floats in DEVICE memory, bit for bit
 * what the sending context's sift_hip_result_device arrays held.  Independent of ctx's own results; returns when the device is
 * done. */
int sift_hip_sparse_unpack(sift_hip_ctx* ctx, const void* dev_records, const void* dev_values, int64_t n_keypoints,
                           void* dev_keypoints, void* dev_descriptors);
/* The image calculate() leaves in the caller's MultiArray: when params.subpixel it is the
 * sigma=1 blurred, 2x nearest-upsampled frame (sift.cpp:20-21); dims of it, then the pixels. */
int sift_hip_image_dims(sift_hip_ctx* ctx, int* w, int* h);
int sift_hip_image_copy(sift_hip_ctx* ctx, int image, float* out);

/* ---- inspection of the last batch (parity tests) ------------------------------------------------ */
/* kind: 0 gaussian (levels 0..D), 1 dog (0..D-1), 2 magnitude, 3 orientation (initial gradient
 * maps, sift.cpp:130-160, only for levels some keypoint scale selects; 0x0 otherwise). */
int sift_hip_level_dims(sift_hip_ctx* ctx, int kind, int octave, int level, int* w, int* h);
int sift_hip_level_copy(sift_hip_ctx* ctx, int image, int kind, int octave, int level, float* out);
float sift_hip_level_scale(sift_hip_ctx* ctx, int kind, int octave, int level);
/* stage: 0 extrema candidates with edge-response flags (sift.cpp:33-34), 1 after first cleanup
 * (:37-42), 2 after orientation assignment (:46), 3 after second cleanup (:49-54), 4 returned. */
int sift_hip_stage_count(sift_hip_ctx* ctx, int image, int stage);
int sift_hip_stage_copy(sift_hip_ctx* ctx, int image, int stage, sift_hip_keypoint* out);

/* ---- sift::alg operators (algorithms.cpp) on single host images, for known-answer tests -------- */
/* Kernel1D::initGaussian taps as the host side computes them; returns radius, -1 on bad sigma. */
int sift_hip_gauss_taps(float sigma, float* taps, int cap);
/* alg::convolveWithGauss (algorithms.cpp:10-22) */
int sift_hip_convolve_with_gauss(sift_hip_ctx* ctx, const float* in, int w, int h, float sigma,
                                 float* out, char* err, int errlen);
/* alg::reduceToNextLevel (:24-36) -> ((w+1)/2, (h+1)/2); alg::increaseToNextLevel (:38-49) -> (2w, 2h) */
int sift_hip_reduce_to_next_level(sift_hip_ctx* ctx, const float* in, int w, int h, float sigma,
                                  float* out, char* err, int errlen);
int sift_hip_increase_to_next_level(sift_hip_ctx* ctx, const float* in, int w, int h, float sigma,
                                    float* out, char* err, int errlen);
/* alg::dog (:52-64) */
int sift_hip_dog(sift_hip_ctx* ctx, const float* lower, const float* higher, int w, int h, float* out);
/* alg::gradientMagnitude / gradientOrientation over a whole level (:108-116, sift.cpp:130-160) */
int sift_hip_gradient(sift_hip_ctx* ctx, const float* in, int w, int h, float* mag, float* ori);
/* Per-point body of Sift::_eliminateEdgeResponses (sift.cpp:295-345) for m points on one DoG
 * triple; flags[i] = 1 if the reference would set filtered. */
int sift_hip_edge_responses(sift_hip_ctx* ctx, const float* dog0, const float* dog1, const float* dog2,
                            int w, int h, const uint16_t* xs, const uint16_t* ys, int m, uint8_t* flags);
/* alg::vertexParabola (:153-178) for m triples (x coords u16, y values f32). */
int sift_hip_vertex_parabola(sift_hip_ctx* ctx, const uint16_t* lnx, const float* lny, const uint16_t* px,
                             const float* py, const uint16_t* rnx, const float* rny, int m, float* out);
/* The cleanup step (sift.cpp:37-42): permutation std::sort(cmpByFilter) applies to n flags;
 * perm[i] = original index of the element ending at position i. */
int sift_hip_sort_by_filter(sift_hip_ctx* ctx, const uint8_t* flags, int n, int32_t* perm);
/* The whole cleanup (sift.cpp:37-42: sort, find first filtered, u16_t size, resize): original
 * indices of the surviving points in their post-sort order.  on_gpu = 1 runs the cleanup kernel
 * (kernels_cleanup.hip; 2 = its global-memory key variant used for very large n), 0 the host's
 * std::sort glue; all must agree. */
int sift_hip_cleanup_survivors(sift_hip_ctx* ctx, const uint8_t* flags, int n, int32_t* survivors,
                               int32_t* count, int on_gpu);

/* ---- a batch over several GPUs of one node, from one process ----------------------------------
 * The reference runs one calculate() on one image (main.cpp:56-57) and keeps no state between images, so a batch shards
 * by image.  A group holds one context per entry of `devices` (a device may be listed more than once), each driven by a
 * persistent host thread of its own; frames are dealt in contiguous blocks (frame i to shard i / ceil(n / shards)).  A shard
 * packs its keypoint lists - records and descriptors, never images - into the sparse wire format on its own GPU and sends
 * them to devices[0] at once on a stream of its own; a gather thread receives all shards' lists there (side by side, one
 * link each) and unpacks them into one array in global image order.  Transport: RCCL point-to-point (ncclSend / ncclRecv over
 * xGMI, one communicator per GPU; librccl.so.1 is opened at run time) when the devices are all different; device-to-device
 * copies when a device is listed twice (RCCL takes one rank per GPU) or RCCL is absent.  There is no collective and no step
 * in which shards wait for each other except that gather.  Status / error behaviour as sift_hip_calculate_batch: the return
 * value and err are those of the first image (in global order) that "threw". */
typedef struct sift_hip_group sift_hip_group;
int sift_hip_group_create(const int* devices, int n_devices, sift_hip_group** out, char* err, int errlen);
void sift_hip_group_destroy(sift_hip_group* group);           /* batches still in flight run to their end first */
int sift_hip_group_shards(sift_hip_group* group);
/* sift_hip_set_option on every shard; and (with no batch in flight)
 *   "gather_wire": 1 (default) lists of shards on other GPUs cross in the sparse wire format and are unpacked on devices[0],
 *                  0 plain arrays, 2 the sparse format for every shard (tests on a one-GPU box);
 *   "gather_transport" (before the first batch): 0 device-to-device copies, 1 (default) RCCL when the devices allow it,
 *                  2 RCCL or the batch fails;
 *   "gather_loopback" (before the first batch; a group of ONE shard): 1 = its lists travel through RCCL to the same rank
 *                  (ncclSend + ncclRecv in one group): the RCCL path on a box with one GPU. */
int sift_hip_group_set_option(sift_hip_group* group, const char* name, int value);
/* 1: the gather runs over RCCL, 0: over copies; `text` (may be NULL) says why.  Makes the communicators if no batch has yet. */
int sift_hip_group_transport(sift_hip_group* group, char* text, int textlen);
/* One batch, start to end: submit + collect. */
int sift_hip_group_calculate(sift_hip_group* group, const float* host_imgs, int n, int w, int h, const sift_hip_params* params,
                             char* err, int errlen);
/* Two batches in flight: submit returns at once (host_imgs must stay valid until the batch has been collected); collect waits
 * for the OLDEST submitted batch, whose results the accessors below then return - until the next-but-one submit, which reuses
 * their buffers.  The gather of batch k runs under the kernels of batch k+1.  A third submit without a collect is refused. */
int sift_hip_group_submit(sift_hip_group* group, const float* host_imgs, int n, int w, int h, const sift_hip_params* params,
                          char* err, int errlen);
int sift_hip_group_collect(sift_hip_group* group, char* err, int errlen);
int sift_hip_group_result_images(sift_hip_group* group);
int sift_hip_group_result_status(sift_hip_group* group, int32_t* status, int cap);
int sift_hip_group_result_counts(sift_hip_group* group, int32_t* counts, int cap);
int64_t sift_hip_group_result_total(sift_hip_group* group);
int sift_hip_group_result_copy(sift_hip_group* group, sift_hip_keypoint* keypoints, float* descriptors);   /* to host or device memory */
int sift_hip_group_result_device(sift_hip_group* group, const void** dev_keypoints, const void** dev_descriptors);   /* on devices[0] */
/* Of the batch collected last: the slowest shard's calculate + pack time, the time from the last shard's report to the lists
 * being in place on devices[0] (transfer + unpack), bytes that crossed devices. */
int sift_hip_group_timing(sift_hip_group* group, double* compute_ms, double* gather_ms, int64_t* gather_bytes);
/* ... and how much of that gather time the collect call itself had to wait for (the rest ran under the next batch). */
int sift_hip_group_gather_exposed(sift_hip_group* group, double* exposed_ms);
/* The library serialises, PER DEVICE, its kernel launches against the runtime calls that crashed beside them (allocations,
 * stream / event creation, the runtime's own copies): sift_amd/csrc/launch_guard.h.  *ms = the time all host threads of the
 * process have spent waiting for such a lock since it started - what a one-process host of several GPUs (sift_hip_group) or
 * of several contexts per GPU pays for that. */
int sift_hip_lock_wait_ms(double* ms);

/* ---- image files and the result overlay (host code, no GPU) ------------------------------------------
 * What /root/reference/main.cpp does around calculate(): vigra::importImage (main.cpp:52-54), cv::imread (:59), the
 * rotated boxes (:60-73) and cv::imwrite (:75).  Read: binary / ASCII PGM and PPM, PNG (every colour type, 1-16 bit,
 * Adam7), JPEG (baseline and progressive Huffman files, 8 bit, greyscale or three components: libjpeg's default decode —
 * islow IDCT, fancy upsampling — restated in sift_amd/csrc/jpeg_decode.cpp, pixel-identical to libjpeg-turbo).  Errors: SIFT_HIP_EPRECONDITION with a text in err. */
int sift_hip_image_info(const char* path, int* w, int* h, int* bands, int* bits, char* err, int errlen);
/* vigra::importImage into a scalar float array: band 0 of a multi-band file (red of RGB / palette), sample values
 * unscaled (0..255; 0..65535 for 16-bit files), x fastest.  `cap` = floats `out` holds (>= w*h). */
int sift_hip_image_read_band0(const char* path, float* out, long long cap, char* err, int errlen);
/* cv::imread(path, CV_LOAD_IMAGE_COLOR): w*h*3 bytes in B,G,R order, grey replicated, alpha dropped, 16 -> 8 bit. */
int sift_hip_image_read_bgr8(const char* path, uint8_t* out, long long cap, char* err, int errlen);
/* cv::imwrite(path + ".png"): 8-bit RGB PNG of a B,G,R buffer. */
int sift_hip_png_write_bgr8(const char* path, const uint8_t* bgr, int w, int h, char* err, int errlen);
/* cv::RotatedRect(center, size, angle).points(): pts = x0 y0 x1 y1 x2 y2 x3 y3 (OpenCV 3.2 order: bottom-left,
 * top-left, top-right, bottom-right for angle 0). */
void sift_hip_rotated_rect_points(float cx, float cy, float width, float height, float angle, float* pts);
/* The box main.cpp:60-67 builds for a keypoint: centre (loc * 2^octave) / (subpixel ? 2 : 1) stored in u16_t (wraps
 * modulo 65536), side = (int)(scale * 10) (cv::Size holds ints), angle = orientation; any output may be NULL. */
void sift_hip_overlay_box(const sift_hip_keypoint* kp, int subpixel, uint16_t* cx, uint16_t* cy, int* side, float* pts);
/* main.cpp:60-73 on a B,G,R image: per keypoint the lines 0-1, 0-3, 2-3, 1-2 of its box, colour Scalar(255,0,0),
 * drawn like cv::line's defaults (corner coordinates rounded to nearest even, clipped, 8-connected, 1 px). */
int sift_hip_overlay_draw(uint8_t* bgr, int w, int h, const sift_hip_keypoint* kps, long long n, int subpixel);

/* ---- measurement ---------------------------------------------------------------------------- */
/* HIP-event timings collected while option "profile" is N > 0 (every N-th batch of the context carries a start / stop event pair
 * on the dispatch packets of the kernels below; no reference counterpart - sift.cpp has no timers).
 * which: 0 = the fused blur family (streaming / LDS-tiled / kept-pixels reduction / the launch of the first two levels),
 *        1 = the two-pass blur fallback, 2 = descriptor_wave_kernel, 3 = extrema_edge_kernel, 4 = the gradient maps' kernel.
 * Returns accumulated milliseconds (the SUM of the launches' durations), launches and ALGORITHMIC bytes (SURVEY.md section 8(d)'s
 * per-unit figures; class 2 reports 0 bytes: its unit is the keypoint, 3.5 KB each, and the caller knows
 * the count) since the last reset. */
int sift_hip_profile_get(sift_hip_ctx* ctx, int which, double* ms, int64_t* launches, double* bytes);
/* Milliseconds during which at least one launch of the class was running (the union of the launches' intervals): equal to the
 * sum above while launches follow one another, smaller when launches of two streams overlap (an octave's top level runs on
 * the side stream). */
int sift_hip_profile_get_busy(sift_hip_ctx* ctx, int which, double* busy_ms);
/* Batches whose launches carried the events since the last reset (what the per-batch figures divide by). */
int sift_hip_profile_batches(sift_hip_ctx* ctx, int64_t* batches);
int sift_hip_profile_reset(sift_hip_ctx* ctx);

const char* sift_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SIFT_HIP_H */
