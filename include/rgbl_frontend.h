/* rgbl_frontend.h — C ABI of librgbl_frontend.so: the MI355X-native RGB-L per-frame front end.
 *
 * This is the drop-in boundary for the hot path of TUMFTM/ORB_SLAM3_RGBL.  The reference has no FFI
 * layer; its boundary is the C++ class API of three classes, and each entry point below names the
 * reference interface it replaces (file:line under /root/reference).  The C++ shims in
 * orb_slam3_rgbl_amd/shim/ re-create those classes (same names, same signatures) on top of this ABI;
 * INTEGRATION.md shows the few lines a maintainer changes.
 *
 * Conventions
 *   - plain C types only; no C++/torch/HIP types in any signature (streams travel as void*).
 *   - every function returns an int status: RGBL_OK (0) or a negative RGBL_ERR_*; the message of the
 *     last failure on the calling thread is available from rgbl_last_error().  Nothing throws or aborts.
 *   - "host" entry points take host pointers, are synchronous and fill caller-owned buffers
 *     (capacity in, count out) — the contract of the reference classes.
 *   - "_device" entry points take device (HBM) pointers, only ENQUEUE work on the handle's stream and
 *     return immediately; results are valid after rgbl_*_sync().  They exist so that batches of
 *     independent frames stay resident in HBM between extract -> depth -> match.
 *   - a handle owns one HIP stream + scratch memory and is NOT re-entrant (like the reference objects:
 *     ORBextractor/DepthModule keep per-call state in members); different handles may be driven from
 *     different threads concurrently.  Matcher entry points are stateless apart from their handle.
 *   - there is NO CPU fallback: if no gfx950 device is usable, create() fails with RGBL_ERR_NO_DEVICE.
 */
#ifndef RGBL_FRONTEND_H
#define RGBL_FRONTEND_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
  RGBL_OK = 0,
  RGBL_ERR_INVALID = -1,   /* bad argument */
  RGBL_ERR_NO_DEVICE = -2, /* no usable HIP device */
  RGBL_ERR_HIP = -3,       /* a HIP runtime call failed */
  RGBL_ERR_CAPACITY = -4,  /* caller buffer too small; counts are still reported */
  RGBL_ERR_OVERFLOW = -5,  /* an internal scratch bound was exceeded (never silently truncated) */
  RGBL_ERR_EMPTY = -6,     /* empty image: mirrors ORBextractor::operator() returning -1 */
  RGBL_ERR_COMM = -7       /* RCCL could not be loaded, or one of its calls failed (rgbl_comm_*, rgbl_gather_*) */
};

const char* rgbl_last_error(void);
/* "hip:gfx950" for the product build.  (The CPU SIMT emulation used by the test-suite reports "emu".) */
const char* rgbl_backend(void);
int rgbl_device_count(void);

/* cv::KeyPoint binary layout (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} rgbl_keypoint;

/* ------------------------------------------------------------------------------------------------
 * ORBextractor            replaces include/ORBextractor.h:49-83, src/ORBextractor.cc:409-469,1086-1195
 * ---------------------------------------------------------------------------------------------- */
typedef struct rgbl_extractor rgbl_extractor;

typedef struct {
  int nfeatures;       /* ORBextractor.nFeatures   */
  float scale_factor;  /* ORBextractor.scaleFactor */
  int nlevels;         /* ORBextractor.nLevels (<= 16) */
  int ini_th_fast;     /* ORBextractor.iniThFAST   */
  int min_th_fast;     /* ORBextractor.minThFAST   */
  int width, height;   /* image size this handle is built for (<= 4096 x 4096) */
  int max_batch;       /* frames per batched call (>= 1) */
} rgbl_extractor_cfg;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.h:49-50).  `device` = HIP device ordinal. */
int rgbl_extractor_create(const rgbl_extractor_cfg* cfg, int device, rgbl_extractor** out);
void rgbl_extractor_destroy(rgbl_extractor* h);

/* Getters GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (ORBextractor.h:61-81) plus mnFeaturesPerLevel and umax; any pointer may be NULL.
 * Arrays hold nlevels entries (umax16: 16). Computed exactly as ORBextractor.cc:414-468 (fp32). */
int rgbl_extractor_tables(const rgbl_extractor* h, float* scale, float* inv_scale, float* sigma2,
                          float* inv_sigma2, int* features_per_level, int* umax16);
/* Upper bound of keypoints one frame can produce (sum over levels of quota + slack). */
int rgbl_extractor_max_keypoints(const rgbl_extractor* h);

/* int ORBextractor::operator()(image, mask [ignored], keypoints, descriptors, vLappingArea)
 * (ORBextractor.h:57-59, ORBextractor.cc:1086-1168).  Host image CV_8UC1 w x h with `stride` bytes per
 * row; writes up to `cap` keypoints and cap x 32 descriptor bytes, *out_n = number of keypoints,
 * *out_mono = the reference's return value (monoIndex).  Empty image -> RGBL_ERR_EMPTY, *out_mono=-1. */
int rgbl_extract(rgbl_extractor* h, const uint8_t* img, int w, int h_, int stride, int lap0, int lap1,
                 rgbl_keypoint* out_kp, uint8_t* out_desc, int cap, int* out_n, int* out_mono);
/* Latency of one frame (optional): the upload of the image, the whole extraction and the copies of its results into the
 * handle's page-locked block are queued and NOT waited for.  The rgbl_extract call that follows with the same image pointer,
 * size, stride and lapping area collects the results (any other extraction call waits for the begun one and drops it).  Work
 * queued on other handles in between - rgbl_depth_prefetch above all - runs next to the extraction, and so does the host.
 * The image must stay unchanged until it is collected. */
int rgbl_extract_begin(rgbl_extractor* h, const uint8_t* img, int w, int h_, int stride, int lap0, int lap1);
/* Drops a begun extraction that will not be collected (waits for it, forgets it).  A begun frame is recognised by its host
 * pointer, size, stride and lapping area alone: call this before freeing a buffer that was begun but never handed to
 * rgbl_extract - an allocator may give the address to the next frame, which would then collect the old frame's results. */
int rgbl_extract_cancel(rgbl_extractor* h);

/* Batched host variant: `batch` images of identical size, image b at imgs + b*frame_stride bytes.
 * Outputs for frame b start at out_kp + b*cap and out_desc + b*cap*32; out_n/out_mono hold batch ints. */
int rgbl_extract_batch(rgbl_extractor* h, const uint8_t* imgs, int batch, int w, int h_, int stride,
                       size_t frame_stride, int lap0, int lap1, rgbl_keypoint* out_kp,
                       uint8_t* out_desc, int cap, int* out_n, int* out_mono);

/* Device-resident batch: d_imgs / d_kp / d_desc / d_n / d_mono are device pointers with the same
 * layout as above (d_n, d_mono: int32[batch]).  Enqueues only; keypoints beyond `cap` are dropped and
 * flagged: rgbl_extractor_sync() then returns RGBL_ERR_CAPACITY. */
int rgbl_extract_batch_device(rgbl_extractor* h, const uint8_t* d_imgs, int batch, int w, int h_,
                              int stride, size_t frame_stride, int lap0, int lap1, rgbl_keypoint* d_kp,
                              uint8_t* d_desc, int cap, int32_t* d_n, int32_t* d_mono);
/* Waits for the handle's stream and reports deferred device-side errors (overflow flags). */
int rgbl_extractor_sync(rgbl_extractor* h);

/* Ingest in front of the extractor: the cv::cvtColor of Tracking::GrabImageRGBL
 * (src/Tracking.cc:1567-1580; also GrabImageStereo / GrabImageMonocular :1469-1556).  8-bit, 3 or 4 interleaved channels
 * -> gray with OpenCV 4.x' fixed-point weights (R 9798, G 19235, B 3735, 15 bits, + 2^14 before the shift).
 * blue_first = 1 is COLOR_BGR2GRAY / COLOR_BGRA2GRAY (settings `Camera.RGB: 0`), 0 is COLOR_RGB2GRAY / COLOR_RGBA2GRAY.
 * Device variant: enqueued on the extractor's stream, so a following rgbl_extract_batch_device() on d_gray is ordered
 * behind it; frame b at d_src + b*src_frame_stride / d_gray + b*gray_frame_stride (bytes). */
int rgbl_cvt_gray_batch_device(rgbl_extractor* h, const uint8_t* d_src, int batch, int channels, int blue_first, int w,
                               int h_, int src_stride, size_t src_frame_stride, uint8_t* d_gray, int gray_stride,
                               size_t gray_frame_stride);
/* cvtColor + operator() in one call on a host colour image (channels 1 = already gray: plain rgbl_extract).
 * out_gray (nullable, gray_stride bytes per row) receives mImGray, which Tracking keeps for the viewer. */
int rgbl_extract_color(rgbl_extractor* h, const uint8_t* img, int channels, int blue_first, int w, int h_, int stride,
                       int lap0, int lap1, rgbl_keypoint* out_kp, uint8_t* out_desc, int cap, int* out_n, int* out_mono,
                       uint8_t* out_gray, int gray_stride);

/* Frame::UndistortKeyPoints / the corner undistortion of Frame::ComputeImageBounds (src/Frame.cc:837-870, 872-900):
 * cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK) - normalise, OpenCV's 5 fixed-point iterations of the inverse
 * Brown-Conrady model in double, re-project with the same K.  K = fx, fy, cx, cy; dist = k1, k2, p1, p2[, k3] (n_dist 4 or 5).
 * SURVEY.md 8(f) row f3.  The reference skips the call when mDistCoef[0] == 0 (the KITTI settings); the shim keeps that test.
 * Host pointers, synchronous: n (x, y) pairs in, n pairs out (in place allowed). */
int rgbl_undistort_points(rgbl_extractor* h, const float* xy, int n, const float K[4], const float* dist, int n_dist, float* out_xy);
/* The keypoints of rgbl_extract_batch_device() (frame b at d_kp + b*cap, d_n counts) -> their undistorted coordinates,
 * d_xy_un[b][i] = (x, y) as 2 floats (cap pairs per frame); enqueued on the handle's stream. */
int rgbl_undistort_keypoints_batch_device(rgbl_extractor* h, const rgbl_keypoint* d_kp, const int32_t* d_n, int batch, int cap,
                                          const float K[4], const float* dist, int n_dist, float* d_xy_un);

/* std::vector<cv::Mat> mvImagePyramid (ORBextractor.h:83; read by Frame::ComputeStereoMatches,
 * Frame.cc:908,998-1013).  Copies level `level` of frame `frame` of the LAST call to host memory.
 * with_border=1 adds the 19-px BORDER_REFLECT_101 frame the reference keeps around every level
 * (ORBextractor.cc:1185-1191): dst is then (w+38) x (h+38).  blurred=1 returns the 7x7 Gaussian
 * working image of ORBextractor.cc:1132-1133 instead (no border). */
int rgbl_extractor_level_size(const rgbl_extractor* h, int level, int* w, int* h_);
int rgbl_extractor_get_level(rgbl_extractor* h, int frame, int level, int blurred, int with_border,
                             uint8_t* dst, int dst_stride);
/* Test/diagnostic access to the stage between FAST and the quad-tree: the candidates handed to
 * DistributeOctTree for (frame, level) in the reference's order, coordinates relative to minBorder
 * (ORBextractor.cc:863-868).  Returns the count through *out_n. */
int rgbl_extractor_get_candidates(rgbl_extractor* h, int frame, int level, rgbl_keypoint* out, int cap,
                                  int* out_n);

/* void Frame::ComputeStereoMatches() (src/Frame.cc:901-1071): for every left keypoint the best
 * right keypoint in its row band (Hamming, strict '<', octave +-1, disparity window from mb / mbf), 11x11 SAD
 * sub-pixel refinement over 11 shifts on the left keypoint's pyramid level, parabola fit, median*2.1 outlier cut.
 * `left` / `right` are the two extractor handles whose LAST extract call processed the left / right image(s): their
 * resident pyramids are read directly (the reference reads mpORBextractor{Left,Right}->mvImagePyramid), so the stereo
 * path needs no pyramid download.  Outputs mvuRight / mvDepth (-1 = no match).
 * Host variant: frame 0 of the last call, host arrays, synchronous.  Device variant: the layout of
 * rgbl_extract_batch_device() for both sides (frame b at + b*cap), enqueued on the left handle's stream. */
int rgbl_stereo_matches(rgbl_extractor* left, rgbl_extractor* right, const rgbl_keypoint* kp_left,
                        const uint8_t* desc_left, int n_left, const rgbl_keypoint* kp_right,
                        const uint8_t* desc_right, int n_right, float mb, float mbf, float* out_uright,
                        float* out_depth);
int rgbl_stereo_matches_batch_device(rgbl_extractor* left, rgbl_extractor* right, int batch,
                                     const rgbl_keypoint* d_kp_left, const uint8_t* d_desc_left,
                                     const int32_t* d_n_left, const rgbl_keypoint* d_kp_right,
                                     const uint8_t* d_desc_right, const int32_t* d_n_right, int cap, float mb,
                                     float mbf, float* d_uright, float* d_depth);

/* Diagnostics: with RGBL_OCTREE_STAMPS set in the environment the quad-tree kernel leaves 16 cycle-counter stamps
 * per (frame, level) workgroup; this copies them out. */
int rgbl_extractor_debug_stamps(rgbl_extractor* h, unsigned long long* out, int count);

/* Self-test on the device: the hand-written instruction wrappers of the kernels (v_mul_u32_u24 with a scalar operand,
 * v_addc_co_u32 with an SGPR mask, global_load_lds_dwordx4 from unaligned sources) against their plain expressions on n
 * random inputs in partially active waves.  RGBL_OK, or RGBL_ERR_HIP with the first mismatch in rgbl_last_error(). */
int rgbl_selftest_wrappers(int device, int n, unsigned seed);

/* Stream control + per-kernel timing (HIP events on the launch stream) for bench.py. */
int rgbl_extractor_set_stream(rgbl_extractor* h, void* hip_stream /* hipStream_t, NULL = own */);
void* rgbl_extractor_stream(rgbl_extractor* h); /* hipStream_t currently used by the handle */
/* the handle's second, internal stream (level-0 FAST / quad-tree and the Gaussian run there next to the resize chain);
 * other handles may queue work behind it with rgbl_*_set_stream */
void* rgbl_extractor_aux_stream(rgbl_extractor* h);
/* Device-side ordering between handles without a host sync: work enqueued on `waiter_stream` after this call
 * starts only when everything enqueued on `signaler_stream` before this call has finished (HIP event). */
int rgbl_stream_wait(void* waiter_stream, void* signaler_stream);
/* Multi-GPU gather, SURVEY.md 8(e): compacts a batch's per-frame results (device arrays as the batch entry points produce
 * them: counts, cap keypoints / descriptors / depths / uRights per frame) into variable-length records of 68 bytes per
 * keypoint (28 B cv::KeyPoint | 32 B descriptor | f32 depth | f32 uRight), frames back to back starting at record
 * `first_record` of d_out (which holds capacity_records records).  d_offsets[f] = first record of frame f,
 * d_offsets[batch] = one past the last; *d_overflow is set to 1 (and the frame skipped) when the records do not fit.
 * Enqueued on `hip_stream`, no host synchronisation.  There is no counterpart in the reference (it runs on one CPU). */
int rgbl_pack_records_device(void* hip_stream, const int32_t* d_n, const rgbl_keypoint* d_kp, const uint8_t* d_desc,
                             const float* d_depth, const float* d_uright, int batch, int cap, long long first_record,
                             long long capacity_records, uint8_t* d_out, long long* d_offsets, int* d_overflow);
/* ---- The gather itself, over RCCL (round 4).  `north_star`: "host stays C++ calling through a thin C-ABI ... RCCL over xGMI
 * only for the final keypoint/descriptor gather".  The reference has no counterpart (one process, one CPU); a multi-GPU host
 * loop in the style of Examples/RGB-L/rgbl_kitti.cc:87-125 (one process per GPU, each tracking its own sequences) calls:
 *
 *   rank 0:  rgbl_comm_unique_id(id);  -> id reaches the other ranks out of band (file, socket, MPI_Bcast, a torch store)
 *   all:     rgbl_comm_create(id, world, rank, device, &comm);            ncclCommInitRank
 *            rgbl_gather_create(comm, device, batch, cap, slots, stream, &g);
 *   per step k (slot = k % slots), behind the step's kernels, nothing waits on the host:
 *            rgbl_gather_pack(g, slot, d_n, d_kp, d_desc, d_depth, d_uright, wait_events, n, done_event);
 *                 pack (rgbl_pack_records_device) + phase 1: ncclAllGather of the per-frame counts of every rank, their copy into
 *                 page-locked host memory, an event behind it
 *   one step later (or all slots at the end: "final" gather):
 *            rgbl_gather_exchange(g, slot);
 *                 phase 2: the host waits for THAT slot's counts only and posts one exact-size ncclSend (non-root) resp. one
 *                 ncclRecv per peer (root) inside one ncclGroup - on MI355X one xGMI link per peer, in parallel
 *   root:    rgbl_gather_sync(g); rgbl_gather_result(g, r, &counts, &d_records, &n) for r = 0 .. world - 1
 *            The root owns TWO receive banks, used alternately: what rgbl_gather_result hands out is overwritten by the SECOND
 *            following rgbl_gather_exchange.  A gather at the end that exchanges several slots in a row must copy every
 *            exchange's records out (rgbl_gather_copy_result, queued on the gather's stream) before the exchange after next.
 *   Errors:  a slot is packed once and exchanged once, in that order (RGBL_ERR_INVALID otherwise - a rank that packed over a
 *            step it never exchanged would be one all-gather ahead of its peers).  An exchange that fails after some transfers
 *            were posted aborts the communicator (ncclCommAbort: the peers' matching calls return with an error instead of
 *            hanging) and the gather handle answers RGBL_ERR_COMM from then on.  rgbl_comm_destroy while gather handles still
 *            use the communicator is deferred to the last rgbl_gather_destroy.
 *
 * Everything is queued on ONE stream the library controls (default: a low-priority stream of its own; or the caller's, e.g. the
 * low-priority stream of the Hamming scan): the collectives do not get a stream of their own the way a framework's process group
 * gives them one.  librccl is loaded lazily (dlopen; an RCCL already mapped into the process - PyTorch's - is reused): a
 * single-GPU process never touches it; rgbl_comm_unique_id / rgbl_comm_create return RGBL_ERR_COMM if it cannot be loaded.
 * comm == NULL in rgbl_gather_create: one rank without RCCL (device copies), the same choreography. */
#define RGBL_COMM_ID_BYTES 128
typedef struct rgbl_comm rgbl_comm;
typedef struct rgbl_gather rgbl_gather;
int rgbl_comm_available(void);                                  /* 1 when an RCCL library can be loaded (loads it) */
int rgbl_comm_unique_id(uint8_t id[RGBL_COMM_ID_BYTES]);        /* ncclGetUniqueId */
int rgbl_comm_create(const uint8_t id[RGBL_COMM_ID_BYTES], int world, int rank, int device, rgbl_comm** out);
void rgbl_comm_destroy(rgbl_comm* c);   /* while gather handles use the communicator it lives on until the last of them is
                                            destroyed; a second call is valid only during that time (afterwards the object is gone) */
int rgbl_comm_info(const rgbl_comm* c, int* world, int* rank, int* device, int* rccl_version);
/* batch frames per step, cap keypoints per frame (the layout of the batch entry points' outputs), slots packed steps that
 * may wait for their exchange (2 = streaming, one step of slack; the number of steps for a gather at the end). */
int rgbl_gather_create(rgbl_comm* comm, int device, int batch, int cap, int slots, void* hip_stream, rgbl_gather** out);
void rgbl_gather_destroy(rgbl_gather* g);
void* rgbl_gather_stream(rgbl_gather* g);
/* one rank only: the root's own records travel through ncclSend / ncclRecv to itself instead of a device copy, so that a
 * one-GPU box exercises the point-to-point path (needs a comm) */
int rgbl_gather_set_loopback(rgbl_gather* g, int enable);
int rgbl_gather_pack(rgbl_gather* g, int slot, const int32_t* d_n, const rgbl_keypoint* d_kp, const uint8_t* d_desc,
                     const float* d_depth, const float* d_uright, void* const* wait_events, int n_wait, void* done_event);
int rgbl_gather_exchange(rgbl_gather* g, int slot);
int rgbl_gather_sync(rgbl_gather* g);   /* waits for the stream; RGBL_ERR_OVERFLOW if a pack did not fit its slot */
/* root, after rgbl_gather_exchange: the records of `rank` of the most recent exchange (valid until the exchange after the
 * next one: two receive banks) and their per-frame counts (host memory, `batch` ints).  The device pointer may be used by
 * work queued on rgbl_gather_stream, or after rgbl_gather_sync. */
int rgbl_gather_result(rgbl_gather* g, int rank, const int32_t** h_counts, const uint8_t** d_records, long long* n_records);
/* convenience: queues a copy of those records into dst (device or host memory, >= n_records * 68 bytes) on the gather's stream */
int rgbl_gather_copy_result(rgbl_gather* g, int rank, void* dst, long long capacity_bytes);
/* The same with explicit, reusable HIP events: record marks a point on a stream, wait makes later work on another
 * stream start only after that point (a never-recorded event does not block). Enables software pipelining across
 * batches: e.g. the extractor may overwrite an output buffer as soon as the event recorded behind its last reader
 * has fired, while the matcher still works on the other buffer. */
int rgbl_event_create(void** out_event);
void rgbl_event_destroy(void* event);
int rgbl_event_record(void* event, void* stream);
int rgbl_event_wait(void* stream, void* event);
/* A stream of the library's device with the lowest (priority < 0), the default (0) or the highest (> 0) scheduling priority,
 * to be handed to the rgbl_*_set_stream calls.  Work that is not on a batch pipeline's critical chain (the Hamming scan of
 * step k next to the extraction of step k + 1) belongs on a low-priority stream: its workgroups fill the slots the chain
 * leaves instead of competing for them (bench.py: +4 %). */
int rgbl_stream_create(void** out_stream, int priority);        /* on the calling thread's current device */
int rgbl_stream_create_on(int device, void** out_stream, int priority); /* on `device` (hipSetDevice first): what multi-GPU hosts call */
void rgbl_stream_destroy(void* stream);
int rgbl_extractor_profile(rgbl_extractor* h, int enable);
/* Returns the number of distinct kernels; fills up to cap entries. names[i] points to static storage. */
int rgbl_extractor_profile_read(rgbl_extractor* h, const char** names, double* total_ms, long* launches,
                                int cap);
/* Every launch's own duration (ms, launch order) of kernel number `kernel` (its index in rgbl_*_profile_read's arrays) since
 * profiling was switched on: returns how many there are, copies the first `cap`.  bench.py builds its per-step median /
 * min / max table from these without synchronising between steps (a host sync per step lets the GPU run dry, and the
 * event bracket of the first launches then measures the host's launch latency). */
int rgbl_extractor_profile_samples(rgbl_extractor* h, int kernel, float* ms, int cap);

/* ------------------------------------------------------------------------------------------------
 * DepthModule             replaces include/DepthModule.h:45-99, src/DepthModule.cc:50-274
 * ---------------------------------------------------------------------------------------------- */
typedef struct rgbl_depth rgbl_depth;

enum { /* DepthModule::UpsamlingMethod (DepthModule.h:34-40) */
  RGBL_UPS_NONE = 0,
  RGBL_UPS_NEAREST_NEIGHBOR_PIXEL = 1,
  RGBL_UPS_AVERAGE_FILTERING = 2,
  RGBL_UPS_INVERSE_DILATION = 3,
  RGBL_UPS_IPBASIC = 5 /* declared by the reference but never implemented: rejected here too */
};

typedef struct {
  float proj[12];      /* LidarProjectionMatrix = K[3x4] * Tr[4x4] (DepthModule.cc:434), row-major */
  float min_dist;      /* LiDAR.min_dist */
  float max_dist;      /* LiDAR.max_dist */
  float mbf;           /* Camera.bf */
  int method;          /* RGBL_UPS_* */
  int kernel_w, kernel_h;  /* inverse-dilation structuring element (<= 9 x 9), anchor = centre */
  uint8_t kernel[81];      /* row-major mask; see rgbl_structuring_element() */
  int avg_kernel_size;     /* LiDAR.MethodAverageFiltering.KernelSize */
  float nn_search_radius;  /* LiDAR.MethodNearestNeighborPixel.SearchDistance */
  int width, height;       /* image size */
  int max_points;          /* capacity for LiDAR points per scan (reference cap: 250000) */
  int max_keypoints;       /* capacity for keypoints per frame */
  int max_batch;
} rgbl_depth_cfg;

/* DepthModule::DepthModule(yaml, sensor) minus the YAML parsing (that stays in the C++ shim). */
int rgbl_depth_create(const rgbl_depth_cfg* cfg, int device, rgbl_depth** out);
void rgbl_depth_destroy(rgbl_depth* h);
/* LidarProjectionMatrix = CameraMatrix * RotationMatrix with OpenCV GEMM arithmetic (host helper). */
void rgbl_projection_matrix(const float K3x4[12], const float Tr4x4[16], float out3x4[12]);
/* cv::getStructuringElement (shape 0 RECT, 1 CROSS, 2 ELLIPSE) and the reference's Diamond tables
 * (shape 3, DepthModule.h:138-161; only kw is used).  out holds kh*kw bytes. */
int rgbl_structuring_element(int shape, int kw, int kh, uint8_t* out);

/* void DepthModule::CalculateDepthFromPcd(mvKeys, mvKeysUn, PointCloud, imwidth, imheight)
 * (DepthModule.h:62, DepthModule.cc:50-79).  cloud = the 4 x n CV_32F matrix of
 * Examples/RGB-L/rgbl_kitti.cc:151-185 (rows x,y,z,1; `ld` floats between rows).  kp_xy = k pairs
 * (pt.x, pt.y) of mvKeys, kpun_x = pt.x of mvKeysUn.  Outputs mvDepth / mvuRight (k floats each) and,
 * if non-NULL, RawDepthMap / ProcessedDepthMap (h*w floats each). */
int rgbl_depth_compute(rgbl_depth* h, const float* cloud, int n, int ld, int w, int h_,
                       const float* kp_xy, const float* kpun_x, int k, float* out_depth,
                       float* out_uright, float* out_raw, float* out_processed);

/* Latency of one frame (optional): the part of CalculateDepthFromPcd that does not depend on the keypoints - upload of the scan,
 * ProjectPointcloudToImage, the up-sampling (DepthModule.cc:57-76 without GetFeatureDepthFromDepthMap) - queued on the handle's
 * stream, not waited for.  Called between rgbl_extract_begin and rgbl_extract it runs next to the extraction; the
 * rgbl_depth_compute / _xyzi call that follows with the same cloud pointer, n and ld then only gathers the keypoints' depths
 * (with out_raw != NULL, or another cloud, it computes everything as usual).  The scan must stay unchanged in between. */
int rgbl_depth_prefetch(rgbl_depth* h, const float* cloud, int n, int ld, int w, int h_);
int rgbl_depth_prefetch_xyzi(rgbl_depth* h, const float* xyzi, int n, int w, int h_);
/* Forgets a prefetched scan that will not be followed by rgbl_depth_compute* (same reason as rgbl_extract_cancel: the scan is
 * recognised by pointer, n and ld).  Every other projecting entry point of the handle (the batch-device ones included)
 * invalidates a prefetch by itself. */
int rgbl_depth_prefetch_cancel(rgbl_depth* h);

/* The same on a scan as it lies in a KITTI velodyne .bin file: n records (x, y, z, reflectance).  Replaces
 * LoadPointcloudBinaryMat's repack to 4 x n (Examples/RGB-L/rgbl_kitti.cc:151-185: reflectance dropped, homogeneous
 * coordinate 1) plus CalculateDepthFromPcd; SURVEY.md 8(f) row f3. */
int rgbl_depth_compute_xyzi(rgbl_depth* h, const float* xyzi, int n, int w, int h_, const float* kp_xy,
                            const float* kpun_x, int k, float* out_depth, float* out_uright, float* out_raw,
                            float* out_processed);

/* Device-resident batch.  d_cloud: batch scans, scan b at d_cloud + b*cloud_stride floats, each 4 x n
 * with leading dimension ld.  Keypoints come straight from rgbl_extract_batch_device(): d_kp (frame b
 * at d_kp + b*kp_cap) and d_n (int32[batch]).  d_kpun_x may be NULL (undistorted == distorted, the
 * KITTI case, Frame.cc:837-845).  Outputs: d_depth/d_uright float[batch*kp_cap]; d_processed (nullable)
 * float[batch*h*w]. */
int rgbl_depth_batch_device(rgbl_depth* h, const float* d_cloud, int batch, int n, int ld,
                            size_t cloud_stride, int w, int h_, const rgbl_keypoint* d_kp,
                            const int32_t* d_n, int kp_cap, const float* d_kpun_x, float* d_depth,
                            float* d_uright, float* d_processed);
/* The two halves of the call above, for overlap: the projection / up-sampling half does not depend on the
 * keypoints, so it can run on the depth handle's stream while the extractor is still busy on its own. */
int rgbl_depth_project_batch_device(rgbl_depth* h, const float* d_cloud, int batch, int n, int ld,
                                    size_t cloud_stride, int w, int h_, float* d_processed);
/* projection half for scans in the .bin layout: scan b = n float4 records at d_xyzi + b*scan_stride floats
 * (16-byte aligned, scan_stride a multiple of 4) */
int rgbl_depth_project_xyzi_batch_device(rgbl_depth* h, const float* d_xyzi, int batch, int n, size_t scan_stride, int w,
                                         int h_, float* d_processed);
int rgbl_depth_gather_batch_device(rgbl_depth* h, int batch, int w, int h_, const rgbl_keypoint* d_kp,
                                   const int32_t* d_n, int kp_cap, const float* d_kpun_x, float* d_depth,
                                   float* d_uright);
int rgbl_depth_sync(rgbl_depth* h);
void* rgbl_depth_stream(rgbl_depth* h); /* hipStream_t currently used by the handle */
int rgbl_depth_set_stream(rgbl_depth* h, void* hip_stream);
/* Sparse up-sampling (off by default).  The reference fills ProcessedDepthMap for every frame (DepthModule.cc:203-251) and
 * then reads it at the keypoints only (DepthModule.cc:82-104); in RGB-L tracking nothing else reads it - its one consumer,
 * FrameDrawer.cc:375, is commented out.  With the switch on, an InverseDilation handle writes the dense map only for calls
 * that ask for it (out_processed / d_processed non-NULL); otherwise the keypoint gather evaluates the dilation at the
 * keypoints' pixels from the projected index map.  mvDepth / mvuRight are bit-identical either way. */
int rgbl_depth_set_sparse(rgbl_depth* h, int enable);
int rgbl_depth_profile(rgbl_depth* h, int enable);
int rgbl_depth_profile_read(rgbl_depth* h, const char** names, double* total_ms, long* launches, int cap);
int rgbl_depth_profile_samples(rgbl_depth* h, int kernel, float* ms, int cap);

/* ------------------------------------------------------------------------------------------------
 * ORBmatcher              replaces include/ORBmatcher.h:43,75-76, src/ORBmatcher.cc:907-1146,2058-2074
 * ---------------------------------------------------------------------------------------------- */
/* What a host-pointer matcher call costs (round 6): the handle owns a device arena and a page-locked block of the same layout;
 * everything a call uploads goes up with ONE request, its result arrays come back with one, the call synchronises once
 * (7 - 25 us beside its kernels).  A frame that is resident on the device (rgbl_device_frame, below) is not uploaded at all. */
typedef struct rgbl_matcher rgbl_matcher;
int rgbl_matcher_create(int device, rgbl_matcher** out);
void rgbl_matcher_destroy(rgbl_matcher* h);
/* Pooled handles for callers that build their matcher per call, as the reference does: `ORBmatcher matcher(0.9, true);`
 * is a function-local object in Tracking::TrackReferenceKeyFrame / TrackWithMotionModel / SearchLocalPoints / Relocalization
 * (Tracking.cc:2525, 2761, 2890, 3424, 3662, 3701) and LocalMapping::CreateNewMapPoints (LocalMapping.cc:412).  acquire gives
 * an idle handle of `device` (or creates the first ones), release parks it; the HIP stream and the device arena live on, so
 * constructing / destroying the drop-in class costs no HIP object.  Handles held at the same time are distinct. */
int rgbl_matcher_acquire(int device, rgbl_matcher** out);
void rgbl_matcher_release(rgbl_matcher* h);
int rgbl_matcher_pool_size(void);  /* idle handles (diagnostics / tests) */
int rgbl_matcher_pool_clear(void); /* destroys the idle handles (streams, device arenas, page-locked blocks): orderly shutdown; returns how many */
int rgbl_matcher_sync(rgbl_matcher* h);
int rgbl_matcher_set_stream(rgbl_matcher* h, void* hip_stream);
void* rgbl_matcher_stream(rgbl_matcher* h);
int rgbl_matcher_profile(rgbl_matcher* h, int enable);
int rgbl_matcher_profile_read(rgbl_matcher* h, const char** names, double* total_ms, long* launches, int cap);
int rgbl_matcher_profile_samples(rgbl_matcher* h, int kernel, float* ms, int cap);

/* static int ORBmatcher::DescriptorDistance(a, b) (ORBmatcher.cc:2058-2074) on two 32-byte rows. Host. */
int rgbl_descriptor_distance(const uint8_t* a, const uint8_t* b);

/* Hamming brute force: for every row of A the best and second-best row of B, selection rule of
 * ORBmatcher.cc:283-304 (strict '<': the first minimum wins; distances start at 256, idx at -1).
 * Host pointers, synchronous.  second_dist may be NULL. */
int rgbl_hamming_bf(rgbl_matcher* h, const uint8_t* desc_a, int na, const uint8_t* desc_b, int nb,
                    int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* Frame::ComputeStereoFishEyeMatches (/root/reference/src/Frame.cc:1256-1296) up to the triangulation:
 * the lapping-area subsets of both images - rows [mono_left, n_left) and [mono_right, n_right), monoLeft / monoRight being
 * what ORBextractor::operator() returned (Frame.cc:512-514) - matched by cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) and
 * Lowe's ratio test `size() >= 2 && best.distance < second.distance * 0.7`.  left_to_right[i] (n_left entries, indices into
 * the FULL right arrays like mvLeftToRightMatch) = the right keypoint that passed, else -1; best_dist / second_dist
 * (nullable, n_left entries, 256 = none) are the two knnMatch distances.  KannalaBrandt8::TriangulateMatches and the
 * `depth > 0.0001f` gate (Frame.cc:1287-1294) stay with the caller, who owns the camera objects.  Host pointers, synchronous. */
int rgbl_stereo_fisheye_matches(rgbl_matcher* h, const uint8_t* desc_left, int n_left, int mono_left, const uint8_t* desc_right,
                                int n_right, int mono_right, int32_t* left_to_right, int32_t* best_dist, int32_t* second_dist);
/* Device batch over frame pairs: descriptors of frame f live at d_desc + f*cap*32 with d_n[f] rows
 * (the layout rgbl_extract_batch_device() writes).  Pair p matches frame pair_a[p] (queries) against
 * frame pair_b[p] (train); outputs for pair p start at p*cap. */
int rgbl_hamming_bf_batch_device(rgbl_matcher* h, const uint8_t* d_desc, const int32_t* d_n, int cap,
                                 const int32_t* d_pair_a, const int32_t* d_pair_b, int n_pairs,
                                 int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist);

/* ---- Frames resident on the device ----------------------------------------------------------------------------------
 * A Frame / KeyFrame is matched many times (Tracking: SearchByProjection, SearchLocalPoints, SearchByBoW on the same
 * CurrentFrame; LocalMapping: SearchForTriangulation / Fuse of one key frame against ~10-20 neighbours), and its per-feature
 * arrays - mDescriptors, mvKeysUn[].pt / .octave, mvuRight (src/Frame.cc:110-125, 302-340; KeyFrame.cc:41-76) - never change
 * after construction.  An rgbl_device_frame holds them in HBM: filled ONCE, either from host arrays (one page-locked block,
 * one copy) or straight from the extractor / depth handles that produced them (device to device - the descriptors and
 * keypoints never cross PCIe a second time), and named by the `device` member of the views below, whose host pointers for
 * these four arrays are then not read.  Optionally the FeatureVector's CSR arrays as well.  A frame object is filled by one
 * thread at a time; once filled it may be read by any number of concurrent matcher calls. */
typedef struct rgbl_device_frame rgbl_device_frame;
int rgbl_device_frame_create(int device, int capacity /* features */, rgbl_device_frame** out);
void rgbl_device_frame_destroy(rgbl_device_frame* f);
/* n <= capacity features from host arrays; uright may be NULL (monocular: all -1).  Synchronous. */
int rgbl_device_frame_upload(rgbl_device_frame* f, int n, const uint8_t* desc, const float* kp_xy, const int32_t* kp_octave,
                             const float* uright);
/* The n keypoints / descriptors frame `frame` of the extractor's LAST rgbl_extract* call left on the device (n = the count
 * that call returned) and, with a depth handle, the mvuRight of ITS last rgbl_depth_compute* call (else -1).  K / dist
 * (nullable) = Frame::UndistortKeyPoints (Frame.cc:837-870): mvKeysUn from mvKeys; nothing is computed when dist is NULL or
 * dist[0] == 0, as upstream.  Enqueued on the extractor's stream behind that extraction; returns without waiting. */
int rgbl_device_frame_capture(rgbl_device_frame* f, rgbl_extractor* ex, int frame, int n, rgbl_depth* depth, const float K[4],
                              const float* dist, int n_dist);
/* mFeatVec as CSR (the node ids stay with the caller: the merge walk of two FeatureVectors is host work).  Synchronous. */
int rgbl_device_frame_set_feature_vector(rgbl_device_frame* f, int n_nodes, const int32_t* node_off, const int32_t* node_feat);
/* Frame::AssignFeaturesToGrid (src/Frame.cc:475-506) for the frame's keypoints, kept with the frame: grid = Frame::mnMinX, mnMinY,
 * mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv.  A projection search whose input names this frame and the same six
 * values skips its grid build (11 - 15 us of a 75 - 90 us call).  Call after upload / capture (they invalidate it).  Synchronous. */
int rgbl_device_frame_set_grid(rgbl_device_frame* f, const float grid[6]);
int rgbl_device_frame_size(const rgbl_device_frame* f);
/* test / debug: the resident arrays back to the host (any pointer may be NULL) */
int rgbl_device_frame_download(rgbl_device_frame* f, uint8_t* desc, float* kp_xy, int32_t* kp_octave, float* uright);

/* int ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse)
 * (ORBmatcher.h:75-76, ORBmatcher.cc:907-1146) on flattened key-frames (mono / stereo pinhole,
 * mpCamera2 == nullptr).  The shim flattens KeyFrame under its mutexes, computes F12 and the epipole
 * once (Pinhole.cpp:109-112, ORBmatcher.cc:913-931) and rebuilds the pair vector from matches12. */
typedef struct {
  int n;                    /* KeyFrame::N */
  const uint8_t* desc;      /* mDescriptors, n x 32 */
  const float* kp_xy;       /* mvKeysUn[i].pt, n x 2 */
  const int32_t* kp_octave; /* mvKeysUn[i].octave */
  const float* kp_angle;    /* mvKeysUn[i].angle */
  const float* uright;      /* mvuRight */
  const uint8_t* has_mappoint; /* GetMapPoint(i) != NULL */
  /* mFeatVec (DBoW2::FeatureVector, a std::map<NodeId, vector<unsigned>>) as CSR, node ids ascending */
  int n_nodes;
  const int32_t* node_id;
  const int32_t* node_off;  /* n_nodes + 1 */
  const int32_t* node_feat; /* feature indices, ascending inside a node */
  /* nullable: the frame's resident copy - desc, kp_xy, kp_octave, uright above are then not read (and node_off / node_feat
   * only by the host-side passes, when the frame holds its FeatureVector: n_nodes must equal what was set) */
  const rgbl_device_frame* device;
} rgbl_keyframe_view;

typedef struct {
  float F12[9];                 /* K1^-T [t12]x R12 K2^-1, row-major */
  float epipole[2];             /* ep = project(T2w * Cw1) in image 2 */
  const float* scale_factors2;  /* pKF2->mvScaleFactors */
  const float* level_sigma2_2;  /* pKF2->mvLevelSigma2 */
  int n_levels;
  int only_stereo;              /* bOnlyStereo */
  int coarse;                   /* bCoarse */
  int check_orientation;        /* mbCheckOrientation (false for LocalMapping.cc:412's matcher) */
} rgbl_triangulation_params;

/* Host pointers, synchronous. matches12 has kf1->n entries (-1 = unmatched); *out_nmatches = return
 * value of the reference function. */
int rgbl_search_triangulation(rgbl_matcher* h, const rgbl_keyframe_view* kf1,
                              const rgbl_keyframe_view* kf2, const rgbl_triangulation_params* prm,
                              int32_t* matches12, int* out_nmatches);
/* int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) (include/ORBmatcher.h:57,
 * src/ORBmatcher.cc:223-425; callers Tracking::TrackReferenceKeyFrame, src/Tracking.cc:2798-2810, and Relocalization).
 * Single-camera frames.  kf: the key-frame (has_mappoint = GetMapPointMatches()[i] != NULL && !isBad(), kp_angle =
 * mvKeysUn[i].angle, FeatureVector as CSR); frame: F.mDescriptors, kp_angle = F.mvKeys[i].angle, F.mFeatVec (the other
 * fields of the view are not read).  match_f (frame->n entries): index of the key-frame feature whose map point ends up
 * in vpMapPointMatches[i], or -1.  Host pointers, synchronous. */
int rgbl_search_by_bow(rgbl_matcher* h, const rgbl_keyframe_view* kf, const rgbl_keyframe_view* frame, float nnratio,
                       int check_orientation, int32_t* match_f, int* out_nmatches);
/* The same for a two-camera frame (F.Nleft != -1: the fisheye rig; src/ORBmatcher.cc:298-326 and 357-386).  frame_n_left =
 * F.Nleft: the frame's features [0, frame_n_left) come from the left camera, the rest from the right one.  A key-frame feature
 * keeps a best / second best among the node's left features and a best among its right ones; if the left best is <= TH_LOW the
 * left one is taken under the ratio test and the right one whenever its own distance is <= TH_LOW (`|| true`, :359: no ratio
 * test) - so one key-frame feature can appear twice in match_f.  kp_angle of both views: the angle of the key point the
 * reference picks for that index (mvKeysUn / mvKeys / mvKeysRight, :335-343, :362-373).  frame_n_left = -1 is
 * rgbl_search_by_bow. */
int rgbl_search_by_bow_rig(rgbl_matcher* h, const rgbl_keyframe_view* kf, const rgbl_keyframe_view* frame, int frame_n_left,
                           float nnratio, int check_orientation, int32_t* match_f, int* out_nmatches);

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (include/ORBmatcher.h:58,
 * src/ORBmatcher.cc:765-905; callers LoopClosing::DetectCommonRegionsFromBoW / DetectAndReffineSim3FromLastKF and
 * Tracking's relocalisation of the multi-map case).  Both views: has_mappoint = GetMapPointMatches()[i] != NULL && !isBad(),
 * kp_angle = mvKeysUn[i].angle, FeatureVector as CSR.  match12 (kf1->n entries): index of the kf2 feature whose map point ends
 * up in vpMatches12[i], or -1.  Host pointers, synchronous. */
int rgbl_search_by_bow_keyframes(rgbl_matcher* h, const rgbl_keyframe_view* kf1, const rgbl_keyframe_view* kf2, float nnratio,
                                 int check_orientation, int32_t* match12, int* out_nmatches);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
 * (include/ORBmatcher.h:48, src/ORBmatcher.cc:1676-1887; callers Tracking::TrackWithMotionModel, src/Tracking.cc:2917-2934):
 * the matcher that runs on every tracked frame.  Single-camera frames (Nleft == -1: RGB-L, RGB-D, stereo, mono pinhole).
 * Includes Frame::GetFeaturesInArea / AssignFeaturesToGrid / PosInGrid (src/Frame.cc:747-825, 475-506).  SURVEY.md 8(f) row f2.
 * CurrentFrame.mvpMapPoints is expected to be all NULL on entry, as Tracking.cc:2913 leaves it. */
typedef struct {
  int n1;                      /* LastFrame.N (< 2^20) */
  const uint8_t* valid1;       /* LastFrame.mvpMapPoints[i] != NULL && !LastFrame.mvbOutlier[i] */
  const float* world_pos1;     /* pMP->GetWorldPos(), 3 floats per feature */
  const uint8_t* mp_desc1;     /* pMP->GetDescriptor(), 32 bytes per feature */
  const uint8_t* mp_observed1; /* pMP->Observations() > 0: such a point blocks the feature it is assigned to (:1745-1747) */
  const int32_t* octave1;      /* LastFrame.mvKeys[i].octave */
  const float* angle1;         /* LastFrame.mvKeysUn[i].angle */
  int n2;                      /* CurrentFrame.N (<= 65535) */
  const float* kp2_xy;         /* CurrentFrame.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* kp2_angle;
  const float* uright2;        /* CurrentFrame.mvuRight */
  const uint8_t* desc2;        /* CurrentFrame.mDescriptors */
  float grid[6];               /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  float Tcw_q[4], Tcw_t[3];    /* CurrentFrame.GetPose(): unit_quaternion() as (x, y, z, w), translation() */
  float Tlw_q[4], Tlw_t[3];    /* LastFrame.GetPose() */
  float K[4];                  /* fx, fy, cx, cy of CurrentFrame.mpCamera (Pinhole::project) */
  float mb, mbf;               /* CurrentFrame.mb, mbf */
  const float* scale_factors;  /* CurrentFrame.mvScaleFactors */
  int n_levels;
  float th;
  int mono;                    /* bMono */
  int check_orientation;       /* mbCheckOrientation */
  const rgbl_device_frame* device2; /* nullable: CurrentFrame resident on the device - kp2_xy, kp2_octave, uright2, desc2 are then not read */
} rgbl_projection_input;
/* Host pointers, synchronous.  match2 has n2 entries: the index of the LastFrame feature whose map point the call leaves
 * in CurrentFrame.mvpMapPoints[i2], or -1.  *out_nmatches = return value of the reference function. */
int rgbl_search_by_projection(rgbl_matcher* h, const rgbl_projection_input* in, int32_t* match2, int* out_nmatches);

/* int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints,
 * const float thFarPoints) (include/ORBmatcher.h:45, src/ORBmatcher.cc:43-213 with RadiusByViewingCos :215-221; caller
 * Tracking::SearchLocalPoints, src/Tracking.cc:3370-3450): the local map points that Frame::isInFrustum found visible are
 * searched around their predicted projection - best / second-best Hamming distance, ratio test inside one pyramid level.
 * Single-camera frames.  SURVEY.md 8(f) row f2. */
/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th,
 * const int ORBdist) (include/ORBmatcher.h:51, src/ORBmatcher.cc:1889-2010; caller Tracking::Relocalization,
 * src/Tracking.cc:3723-3752).  Single-camera frames.  The caller (shim) evaluates what needs the MapPoint objects:
 * valid1 and MapPoint::PredictScale (src/MapPoint.cc:531-546, a logf and a ceil); projection, window search, the greedy
 * assignment in key-frame index order and the rotation histogram are done here. */
typedef struct {
  int n1;                      /* pKF->GetMapPointMatches().size() (< 2^20) */
  const uint8_t* valid1;       /* pMP != NULL && !pMP->isBad() && !sAlreadyFound.count(pMP) &&
                                  minDistance <= |x3Dw - Ow| <= maxDistance (GetMin/MaxDistanceInvariance) */
  const float* world_pos1;     /* pMP->GetWorldPos(), 3 floats per point */
  const uint8_t* mp_desc1;     /* pMP->GetDescriptor(), 32 bytes per point */
  const int32_t* level1;       /* pMP->PredictScale(dist3D, &CurrentFrame) */
  const float* angle1;         /* pKF->mvKeysUn[i].angle */
  int n2;                      /* CurrentFrame.N (<= 65535) */
  const float* kp2_xy;         /* CurrentFrame.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* kp2_angle;
  const uint8_t* desc2;        /* CurrentFrame.mDescriptors */
  const uint8_t* occupied2;    /* CurrentFrame.mvpMapPoints[i] != NULL on entry (nullable: none) */
  float grid[6];               /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  float Tcw_q[4], Tcw_t[3];    /* CurrentFrame.GetPose(): unit_quaternion() as (x, y, z, w), translation() */
  float K[4];                  /* fx, fy, cx, cy of CurrentFrame.mpCamera (Pinhole::project) */
  const float* scale_factors;  /* CurrentFrame.mvScaleFactors */
  int n_levels;
  float th;
  int orb_dist;                /* ORBdist, 0..255 */
  int check_orientation;       /* mbCheckOrientation */
  const rgbl_device_frame* device2; /* nullable: CurrentFrame resident on the device - kp2_xy, kp2_octave, uright2, desc2 are then not read */
} rgbl_keyframe_projection_input;
/* Host pointers, synchronous.  match2 (n2 entries): index of the key-frame feature whose map point the call stores in
 * CurrentFrame.mvpMapPoints[i2], or -1 (entry left as it was).  *out_nmatches = return value of the reference function. */
int rgbl_search_by_projection_keyframe(rgbl_matcher* h, const rgbl_keyframe_projection_input* in, int32_t* match2,
                                       int* out_nmatches);

/* The per-point search of the Sim3-based loop-closing matchers: ORBmatcher::Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const
 * vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint) (include/ORBmatcher.h:86, src/ORBmatcher.cc:1340-1455)
 * and both directions of SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (include/ORBmatcher.h:79, src/ORBmatcher.cc:1457-1674).
 * The caller (shim) applies its Sophus objects and evaluates the tests on the MapPoint objects; from the camera-frame
 * coordinates on - positive depth, projection, KeyFrame::IsInImage, the window of radius th * scale[level], octaves level - 1
 * ... level, the smallest Hamming distance - everything is done here, every point on its own. */
typedef struct {
  int n1;
  const uint8_t* valid1;       /* the point passed the caller's tests (NULL / bad / already matched / invariance range / viewing angle) */
  const float* cam_pos1;       /* the point in the key frame's camera frame (Tcw * p3Dw, resp. S21 * (T1w * p3Dw)), 3 floats */
  const uint8_t* mp_desc1;     /* pMP->GetDescriptor() */
  const int32_t* level1;       /* pMP->PredictScale(dist3D, pKF) */
  int n2;                      /* pKF->N (<= 65535) */
  const float* kp2_xy;         /* pKF->mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const uint8_t* desc2;
  float grid[6];               /* pKF->mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  float K[4];                  /* fx, fy, cx, cy */
  const float* scale_factors;
  int n_levels;
  float th;
  int proj_form;               /* 0: Pinhole::project (fx x / z + cx) as in Fuse; 1: invz = 1.0 / z (double), fx (x invz) + cx as in
                                  SearchBySim3; 2 (rgbl_search_by_projection_sim3 only): invz = 1 / z in float */
  int max_dist;                /* TH_LOW (Fuse) / TH_HIGH (SearchBySim3) */
  const rgbl_device_frame* device2; /* nullable: pKF resident on the device - kp2_xy, kp2_octave, uright2, desc2 are then not read */
} rgbl_project_search_input;
/* Host pointers, synchronous.  best_idx[i] = key-frame feature with the smallest distance (<= max_dist) or -1; best_dist nullable. */
int rgbl_project_search(rgbl_matcher* h, const rgbl_project_search_input* in, int32_t* best_idx, int32_t* best_dist);

/* int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints,
 * vector<MapPoint*>& vpMatched, int th, float ratioHamming) (include/ORBmatcher.h:61, src/ORBmatcher.cc:427-532; proj_form 0) and
 * the overload that also fills vpMatchedKF (include/ORBmatcher.h:65, src/ORBmatcher.cc:534-646; proj_form 2: invz = 1 / z in
 * float), the matchers of LoopClosing::FindMatchesByProjection.  Input as for rgbl_project_search (camera-frame points, the
 * caller's tests folded into valid1, max_dist = floor(TH_LOW * ratioHamming)); matched2[i] != 0: vpMatched[i] != NULL on entry
 * (nullable).  Points are taken in index order and a matched feature is skipped by the points after it.
 * match2 (n2 entries): index of the point stored in vpMatched[i2] by this call, or -1.  Host pointers, synchronous. */
int rgbl_search_by_projection_sim3(rgbl_matcher* h, const rgbl_project_search_input* in, const uint8_t* matched2, int32_t* match2,
                                   int* out_nmatches);

/* void MapPoint::ComputeDistinctiveDescriptors() (src/MapPoint.cc:329-403; called after every new observation / fusion, e.g.
 * LocalMapping.cc:333,691, Tracking.cc:2437) for a batch of map points: the N x N ORBmatcher::DescriptorDistance table of a point's
 * observed descriptors, and the row with the least median.  desc: the rows vDescriptors collects, point p = rows off[p] ..
 * off[p+1]) in the order the reference pushes them (off[0] = 0, at most 65535 per point).  best[p] = BestIdx inside the point's
 * own list (the descriptor to clone into mDescriptor), -1 for a point with no descriptor.  Host pointers, synchronous. */
int rgbl_distinctive_descriptors(rgbl_matcher* h, const uint8_t* desc, const int32_t* off, int n_points, int32_t* best);

/* The search inside int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th, const bool bRight)
 * (include/ORBmatcher.h:83, src/ORBmatcher.cc:1148-1338, bRight = false; caller LocalMapping::SearchInNeighbors,
 * src/LocalMapping.cc:737-879, twice per neighbour key frame).  Every map point finds its best feature independently of
 * the others; what the loop then does with a match (MapPoint::Replace / AddObservation, KeyFrame::AddMapPoint) mutates the
 * caller's objects and stays in the shim, as do the tests that need the MapPoint object. */
typedef struct {
  int n1;                         /* vpMapPoints.size() */
  const uint8_t* valid1;          /* pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF) && minDistance <= dist3D <= maxDistance &&
                                     !(PO.dot(Pn) < 0.5 * dist3D) */
  const float* world_pos1;        /* pMP->GetWorldPos() */
  const uint8_t* mp_desc1;        /* pMP->GetDescriptor() */
  const int32_t* level1;          /* pMP->PredictScale(dist3D, pKF) */
  int n2;                         /* pKF->N (<= 65535) */
  const float* kp2_xy;            /* pKF->mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* uright2;           /* pKF->mvuRight */
  const uint8_t* desc2;           /* pKF->mDescriptors */
  float grid[6];                  /* pKF->mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  float Tcw_q[4], Tcw_t[3];       /* pKF->GetPose() */
  float K[4];                     /* fx, fy, cx, cy (pKF->mpCamera->project) */
  float bf;                       /* pKF->mbf */
  const float* scale_factors;     /* pKF->mvScaleFactors */
  const float* inv_level_sigma2;  /* pKF->mvInvLevelSigma2 */
  int n_levels;
  float th;
  const rgbl_device_frame* device2; /* nullable: pKF resident on the device - kp2_xy, kp2_octave, uright2, desc2 are then not read */
} rgbl_fuse_input;
/* Host pointers, synchronous.  best_idx[i] = the key-frame feature point i would be fused with (bestDist <= TH_LOW), or -1;
 * best_dist (nullable) = bestDist, 256 when the point had no candidate. */
int rgbl_fuse_search(rgbl_matcher* h, const rgbl_fuse_input* in, int32_t* best_idx, int32_t* best_dist);

typedef struct {
  int n1;                      /* vpMapPoints.size() (< 2^20) */
  const uint8_t* valid1;       /* pMP->mbTrackInView && !(bFarPoints && pMP->mTrackDepth > thFarPoints) && !pMP->isBad() */
  const float* proj1;          /* pMP->mTrackProjX, mTrackProjY, mTrackProjXR: 3 floats per point */
  const int32_t* level1;       /* pMP->mnTrackScaleLevel */
  const float* view_cos1;      /* pMP->mTrackViewCos */
  const uint8_t* mp_desc1;     /* pMP->GetDescriptor(), 32 bytes per point */
  const uint8_t* mp_observed1; /* pMP->Observations() > 0 */
  int n2;                      /* F.N (<= 65535) */
  const float* kp2_xy;         /* F.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* uright2;        /* F.mvuRight */
  const uint8_t* desc2;        /* F.mDescriptors */
  const uint8_t* blocked2;     /* F.mvpMapPoints[i] != NULL && ->Observations() > 0 on entry (nullable: none) */
  float grid[6];               /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  const float* scale_factors;  /* F.mvScaleFactors */
  int n_levels;
  float th;
  float nnratio;               /* mfNNratio */
  const rgbl_device_frame* device2; /* nullable: F resident on the device - kp2_xy, kp2_octave, uright2, desc2 are then not read */
} rgbl_local_points_input;
/* Host pointers, synchronous.  match2 (n2 entries): index of the map point the call stores in F.mvpMapPoints[i2], or -1
 * (entry left as it was).  *out_nmatches = return value of the reference function. */
int rgbl_search_local_points(rgbl_matcher* h, const rgbl_local_points_input* in, int32_t* match2, int* out_nmatches);

/* ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12,
 * int windowSize)    /root/reference/include/ORBmatcher.h:72, src/ORBmatcher.cc:648-763 (Tracking::MonocularInitialization,
 * src/Tracking.cc:2526; not on the RGB-L / stereo path - built so that every search routine of ORBmatcher has a device form).
 * Level-0 features of F1 look for their best / second-best F2 feature of level 0 inside a window around vbPrevMatched; a
 * feature takes a candidate over from an earlier one when it is strictly closer (vMatchedDistance), rotation histogram. */
typedef struct {
  int n1;                     /* F1.mvKeysUn.size() (< 2^20) */
  const int32_t* kp1_octave;  /* F1.mvKeysUn[i].octave (>= 0) */
  const float* kp1_angle;     /* F1.mvKeysUn[i].angle */
  const uint8_t* desc1;       /* F1.mDescriptors */
  int n2;                     /* F2.mvKeysUn.size() (<= 65535) */
  const float* kp2_xy;        /* F2.mvKeysUn[i].pt */
  const int32_t* kp2_octave;
  const float* kp2_angle;
  const uint8_t* desc2;       /* F2.mDescriptors */
  float grid[6];              /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  int window_size;
  float nnratio;              /* mfNNratio */
  int check_orientation;      /* mbCheckOrientation */
} rgbl_initialization_input;
/* Host pointers, synchronous.  prev_matched (2 floats per F1 feature) = vbPrevMatched, read as the window centres and
 * updated for the matched features like the reference does; matches12 (n1 entries) = vnMatches12; *out_nmatches = the
 * return value. */
int rgbl_search_for_initialization(rgbl_matcher* h, const rgbl_initialization_input* in, float* prev_matched, int32_t* matches12,
                                   int* out_nmatches);

/* ------------------------------------------------------------------------------------------------
 * ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>)    SURVEY.md 8(f) row f4
 *   replaces the per-feature descent of Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:828-835,
 *   src/KeyFrame.cc:98-107: mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)),
 *   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1255, FORB.cpp:79-98.  The tree lives on the device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct rgbl_vocabulary rgbl_vocabulary;
/* ORBVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1425; System.cc:115): the ORBvoc.txt format. */
int rgbl_vocabulary_load_text(const char* path, int device, rgbl_vocabulary** out);
/* The same from flat arrays: node 0 = root, children of node i = child[child_off[i] .. child_off[i+1]) in file order,
 * 32-byte node descriptors, Node::weight, Node::word_id (leaves). */
int rgbl_vocabulary_create(int n_nodes, int L, const int32_t* child_off, const int32_t* child, const uint8_t* desc,
                           const double* weight, const int32_t* word_id, int device, rgbl_vocabulary** out);
void rgbl_vocabulary_destroy(rgbl_vocabulary* v);
int rgbl_vocabulary_info(const rgbl_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words);
/* transform(features, BowVector, FeatureVector, levelsup).  Host pointers, synchronous.  BowVector as ascending word ids +
 * values (TF-IDF, L1-normalised doubles), FeatureVector as ascending node ids + CSR offsets (n_nodes + 1) + feature
 * indices (ascending inside a node).  RGBL_ERR_CAPACITY when cap_words / cap_nodes (n is always enough) are too small. */
int rgbl_bow_transform(rgbl_vocabulary* v, const uint8_t* desc, int n, int levelsup, uint32_t* word_id, double* word_val,
                       int cap_words, int* n_words, uint32_t* node_id, int32_t* node_off, uint32_t* node_feat, int cap_nodes,
                       int* n_nodes);
/* The same on a frame whose descriptors are resident (rgbl_device_frame): nothing is uploaded. */
int rgbl_bow_transform_frame(rgbl_vocabulary* v, const rgbl_device_frame* frame, int levelsup, uint32_t* word_id, double* word_val,
                             int cap_words, int* n_words, uint32_t* node_id, int32_t* node_off, uint32_t* node_feat, int cap_nodes,
                             int* n_nodes);
/* Device-resident batch, descent only: the descriptors of rgbl_extract_batch_device() (frame b at d_desc + b*cap*32, d_n
 * counts) -> per feature word id, word weight, node id at level L - levelsup.  Enqueued on hip_stream (NULL: own stream). */
int rgbl_bow_descend_batch_device(rgbl_vocabulary* v, void* hip_stream, const uint8_t* d_desc, const int32_t* d_n, int batch,
                                  int cap, int levelsup, int32_t* d_word, double* d_weight, int32_t* d_node);

/* F12 with the reference's fp32 evaluation order (Pinhole.cpp:109-112); K = {fx, fy, cx, cy}. Host. */
void rgbl_fundamental(const float K1[4], const float K2[4], const float R12[9], const float t12[3],
                      float F12[9]);

/* ------------------------------------------------------------------------------------------------
 * Environment switches.  Every one of them is read ONCE, when the handle it concerns is created (never on a launch path:
 * the entry points are called from the three SLAM threads), and is a tuning / test aid - the defaults are what is measured
 * and shipped.  Results are bit-identical under all of them (parity_checks.check_switches: the batch and the single-frame
 * extraction and the Hamming scan under every switch that changes a launch path, on the emulator and on the MI355X).
 *
 *   rgbl_extractor_create
 *     RGBL_SPLIT_PYR=k      batches: the pyramid levels k .. L-1 and their FAST cells leave the main launch chain for the auxiliary
 *                           stream (default L / 2 from 6 levels on; 0 = off)
 *     RGBL_LEVEL_SPLIT=k    single frames: the levels 1 .. k-1 get a stream of their own (default 3; 0 = off)
 *     RGBL_FAST_BS=64|128   waves per FAST detection cell (default: 64 for batches of >= 8 frames, 128 for single frames)
 *     RGBL_COMPACT=0|1      k_compact_cells never / always (default: batches of >= 8 frames)
 *     RGBL_DENSE=0          quad-tree reads the cells' own slots instead of a dense per-level candidate list
 *     RGBL_OCTREE_NCAP=0|512|2048   quad-tree node capacity in LDS (0 = the key-moving kernel on global lists; default by nFeatures)
 *     RGBL_OCTREE_WG=256|512        quad-tree workgroup width (default by batch size)
 *     RGBL_OCTREE_LDSKEYS=0         single frames: candidate lists stay in global memory
 *     RGBL_OCTREE_HIST=0            breadth-first rounds as passes over the keys instead of on the per-cell count pyramid
 *     RGBL_OCTREE_STAMPS=1          the quad-tree kernel leaves phase time stamps (rgbl_extractor_debug_stamps)
 *     RGBL_GAUSS_BS=256     four-wave workgroups for the Gaussian (default two waves)
 *     RGBL_XCD_MAP=0        plain (items, frames) grids instead of the XCD-aware (8, items, frames / 8) mapping (also rgbl_depth_create)
 *     RGBL_GRAPH=0          host-pointer extraction without hipGraph replay
 *   rgbl_depth_create
 *     RGBL_DEPTH_MAX_GEN=n  generations of the index map before it is cleared (tests of the wrap-around)
 *   rgbl_matcher_create
 *     RGBL_BF_MFMA=0|i8     Hamming scan on the VALU (popcount) / on v_mfma_i32_32x32x32_i8 (default: block-scaled FP4 instruction)
 *     RGBL_BF_SPLIT=0       one pair per call without train-set slices
 *   first rgbl_comm_* call
 *     RGBL_RCCL_LIB=path    librccl to dlopen when none is mapped into the process yet
 * ---------------------------------------------------------------------------------------------- */

#ifdef __cplusplus
}
#endif
#endif /* RGBL_FRONTEND_H */
