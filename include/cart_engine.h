/*
 * cart_engine.h -- C ABI of the MI355X (gfx950) dense-stereo engine.
 *
 * This is the drop-in boundary for CART-SLAM's per-frame stereo hot path.  Every
 * entry point names the reference interface it replaces (paths relative to the
 * LorgeN/CART-SLAM tree).  Conventions:
 *   - plain C types and one opaque handle; no C++/torch/OpenCV types;
 *   - image pointers are DEVICE pointers, row-pitched (`*_step` in BYTES, like
 *     cv::cuda::GpuMat::step); the caller owns every image buffer, the engine owns
 *     only its workspaces (census maps, cost slabs, WTA maps) sized at create time;
 *   - every call returns 0 on success, non-zero on failure, never throws and never
 *     exits (the reference's CUDA_SAFE_CALL -> exit(), include/utils/cuda.cuh:193-201,
 *     is deliberately NOT replicated); cart_last_error() gives the message;
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  The module adapter synchronises to mimic
 *     cv::cuda::Stream::waitForCompletion() (src/modules/disparity/disparity.cu:77);
 *   - thread-safe: one engine may be entered concurrently from many host threads
 *     (the reference enters one module object for up to 12 frames at once,
 *     include/cartslam.hpp:4-5); each call leases `n_frames` workspace slots.
 */
#ifndef CART_ENGINE_H
#define CART_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CART_DISPARITY_INVALID (-32768) /* include/modules/disparity.hpp:17 */

/* include/modules/planeseg.hpp:37-41 */
enum { CART_PLANE_HORIZONTAL = 0, CART_PLANE_VERTICAL = 1, CART_PLANE_UNKNOWN = 2 };

typedef struct cart_engine cart_engine;

/* Constructor arguments of cart::ImageDisparityModule (include/modules/disparity.hpp:26-34,
 * JSON keys src/cartconfig.cpp:144-152) plus the cv::cuda::createStereoSGM parameters the
 * reference leaves at OpenCV's defaults (P1, P2, mode -> paths). */
typedef struct {
    int device_id;
    int width, height;        /* DataSource::getImageSize(), include/datasource.hpp:75 */
    int min_disparity;        /* "min_disparity", default 4; 0..64 supported */
    int num_disparities;      /* "num_disparities": 64 | 128 | 256; 0 (with paths 0) = geometry-only engine for the
                                 post-SGM entry points: no SGM workspaces, cart_compute_disparity* fail */
    int paths;                /* 4 (MODE_HH4) | 8 (MODE_HH) */
    int p1, p2;               /* 10, 120 ; 31 + p2 must fit u8 */
    int uniqueness_ratio;     /* disparity.hpp:32 -> 12 */
    int smoothing_radius;     /* "smoothing_radius", default -1 (off); <= 8 */
    int smoothing_iterations; /* "smoothing_iterations", default 5 */
    int max_inflight;         /* workspace slots = frames that may be in flight / batched */
} cart_engine_params;

/* include/modules/planeseg.hpp:25-34 (PlaneParameters) */
typedef struct {
    int horizontal_min, horizontal_max; /* horizontalRange.first / .second */
    int vertical_min, vertical_max;     /* verticalRange.first / .second */
    int horizontal_center, vertical_center;
} cart_plane_params;

/* Fills *p with the reference's defaults (cartconfig.cpp:144-152, disparity.hpp:26-34). */
void cart_engine_default_params(cart_engine_params *p);

/* replaces: ImageDisparityModule ctor + cv::cuda::createStereoSGM (disparity.hpp:26-34) */
int cart_engine_create(const cart_engine_params *params, cart_engine **out);
void cart_engine_destroy(cart_engine *engine);

/* Launch plans of the SGM core.  Every plan produces the same bits; they differ in which path slabs exist in HBM.
 *   SLABS     all P path slabs are written by the aggregation launch and read by the WTA launch (2*P*D bytes / pixel);
 *   FUSED_UP  the "up" path is computed inside the WTA sweep and never stored (2*(P-1)*D bytes / pixel).
 * AUTO picks per launch from the measured table in DESIGN.md section 4.  Options are plain integers so that the
 * boundary stays C; nothing in the engine reads the environment.
 */
enum { CART_PLAN_AUTO = -1, CART_PLAN_SLABS = 0, CART_PLAN_FUSED_UP = 1 };
enum {
    CART_OPT_PLAN = 0,            /* CART_PLAN_*; default AUTO */
    CART_OPT_PLAN_MIN_FRAMES = 1, /* with a forced plan: launches of fewer frames take SLABS (default 1) */
    CART_OPT_CHUNK_FRAMES = 2,    /* frames per launch sequence inside one batched call (default 16, 1..64) */
    /* The three choices that are open upstream (cv::cuda::StereoSGM is un-vendored and un-versioned in the reference:
     * oracle/cart_oracle.h, NOTEs at S7 and S8).  Default 0 = the oracle's spec.  Whoever runs tools/ref_pin against the
     * reference's OpenCV flips the one that its outputs ask for; both forms are held bit-exact against the oracle's
     * variants by tests/test_gpu_parity.py::test_spec_variants.  These change results, by design; nothing else does. */
    CART_OPT_SPEC_S8_ZERO_INVALID = 3,      /* 1: the LR check also invalidates pixels whose integer disparity is 0 (older libSGM's `d <= 0`) */
    CART_OPT_SPEC_S7_REPLICATE_BORDER = 4,  /* 1: the 3x3 medians filter the one-pixel image border over a replicated border instead of passing it through */
    CART_OPT_SPEC_S5_TOP2 = 5               /* 1: uniqueness tests the second-best (cost, d) only -- the top-2 wording of SURVEY.md 8a-4(4) -- instead of every
                                               disparity (the libSGM form, oracle S5); such an engine always takes plan SLABS */
};
int cart_engine_set_option(cart_engine *engine, int option, int value);
int cart_engine_get_option(cart_engine *engine, int option, int *value);
/* What a batched call of n_frames will do: frames per launch sequence, the plan of a full launch, and the number of
 * path slabs that plan materialises (bench.py prices its roofline line from this instead of re-deriving it). */
typedef struct {
    int frames_per_launch;
    int plan;             /* CART_PLAN_SLABS | FUSED_UP */
    int slabs_written;    /* u8 slabs of D bytes per pixel written (and read once) per frame */
} cart_launch_plan;
int cart_engine_describe_plan(cart_engine *engine, int n_frames, cart_launch_plan *out);

/* Placement tuning of the cost-slab workspace (no reference counterpart; OPTIONAL and opt-in: nothing calls it unless the caller asks).
 * On MI355X the time of the two slab-bound launches depends on WHICH physical memory backs the slabs: the same kernels run in one of two
 * modes per allocation (aggregation 1.41-1.43 or 1.53-1.55 ms, WTA 1.15 or 1.27 ms per 16 pairs at 1242x375 D=128 P=8;
 * profiles/r03_alloc.txt: the L2's write requests to the fabric stall 20-30x as often in the slow mode, TLB misses and clock are the same;
 * allocations above 8 GiB are always slow, which is why the engine cuts its slab workspace into groups of slots, each one plain hipMalloc
 * of at most 8 GiB - 64 MiB).
 * The call works per UNIT = the slot groups behind slots [k n, (k+1) n) of an `n_frames` call (n = min(n_frames, frames per launch)),
 * for the first (at most four) such ranges: it times the aggregation + WTA launches of n frames on the unit's current allocations, then on
 * up to `max_tries - 1` freshly allocated sets, keeps the fastest and frees the others (a candidate is compared with the kept set RE-TIMED right after it and replaces it when it is
 * 1.5 % faster: the clock drifts by more than the modes differ over a search).  A unit's search stops (cart_placement_report::stop_reason)
 *   FAST_FOUND  once the kept set is 7 % faster than the slowest one seen (both launches fast against both slow);
 *   UNIFORM     once six sets have been timed and the slowest is within 1.5 % of the kept one: the pool offers one kind of placement only
 *               (about one fresh box in ten; 64 tries bought 1.4 % on such a box) -- the best seen is kept, set-up stays under 2 s;
 *   TRIES / TIME / MEMORY  out of tries, out of the unit's time share ((0.25 s per allowed try + 1 s per 20 GB of workspace) / units),
 *               or no room for another candidate.
 * `mode` says what unit 0 -- where a caller with one call in flight lives -- ended on, RELATIVE to what the search saw (the engine knows no absolute
 * level): FAST = the kept set is at least 5.5 % under the slowest seen, i.e. a set with a launch in its slow mode was met and avoided; MIXED = the sets
 * seen differ by 1.5-5.5 % and the fastest is kept (no both-slow set was met to compare with: the kept one may well have both launches fast); UNIFORM =
 * six sets within 1.5 % of each other (all fast or all slow: the stage times tell which -- at 1242x375 D=128 P=8 the aggregation launch takes ~1.45 ms in
 * its fast mode and ~1.55 in its slow one); UNKNOWN = one try, nothing to compare with.  The ratios hold for a probe on a warmed-up GPU (call it after
 * a few real calls, as bench.py does; straight after engine creation the levels lie 12-13 % apart).
 * TRANSIENT FOOTPRINT: candidates that lost stay allocated while the search goes on (freed at once, the allocator would hand the same
 * pages back); at no time does the call hold more than `max_extra_bytes` beyond the engine's own workspace -- 0 selects two units' worth
 * (one unit = the groups of one n-frame call: 7.6 GB at 1242x375 D=128 P=8 with n = 16), SIZE_MAX lifts the cap (the search then stops
 * when the next candidate would not leave 4 GiB of device memory free; that check is not atomic across processes: pass a finite cap when
 * several processes share a GPU).  With a cap below one unit the call measures and returns without trying anything.  Peak device memory
 * of the process during the call = workspace + min(max_extra_bytes, (max_tries - 1) x unit); after it, the workspace alone.
 * The engine must be idle; results do not change (every placement gives the same bits).  `report` may be NULL. */
enum { CART_PLACE_MODE_UNKNOWN = 0, CART_PLACE_MODE_FAST = 1, CART_PLACE_MODE_MIXED = 2, CART_PLACE_MODE_UNIFORM = 3 };
enum { CART_PLACE_STOP_NOTHING_TO_DO = 0, CART_PLACE_STOP_FAST_FOUND = 1, CART_PLACE_STOP_UNIFORM = 2, CART_PLACE_STOP_TRIES = 3,
       CART_PLACE_STOP_TIME = 4, CART_PLACE_STOP_MEMORY = 5 };
typedef struct {
    float ms_first, ms_kept;                 /* launch-pair time (aggregation + WTA of n frames), mean over the probed units, before / after */
    float ms_fastest_seen, ms_slowest_seen;  /* unit 0: the kept placement and the slowest one timed */
    float seconds;                           /* wall time of the call */
    int units, candidates;                   /* units probed; placements timed in all (the initial ones included) */
    int mode, stop_reason;                   /* CART_PLACE_MODE_* / CART_PLACE_STOP_* of unit 0 */
} cart_placement_report;
int cart_engine_tune_placement(cart_engine *engine, int n_frames, int max_tries, size_t max_extra_bytes, cart_placement_report *report);

/* Message of the last failed call made by THIS thread on `engine` (or of a failed
 * create when engine == NULL).  Never NULL. */
const char *cart_last_error(const cart_engine *engine);

/* replaces: ImageDisparityModule::runInternal (src/modules/disparity/disparity.cu:49-80):
 * cvtColor x2 (:66-67) -> StereoSGM::compute (:71) -> disparity::interpolate (:73-75,
 * src/modules/disparity/interpolation.cu:85-99).  channels = 1 (gray) or 3 (BGR8).
 * out: CV_16SC1-shaped, disparity x16, invalid pixels as the reference produces them. */
int cart_compute_disparity(cart_engine *engine, const uint8_t *left, size_t left_step,
                           const uint8_t *right, size_t right_step, int channels,
                           int16_t *out, size_t out_step, void *stream);

/* Batched-frame mode (north-star config 5): frame f of each array starts at
 * base + f * <frame_stride> BYTES.  n_frames <= max_inflight. */
int cart_compute_disparity_batch(cart_engine *engine, int n_frames,
                                 const uint8_t *left, size_t left_step, size_t left_frame_stride,
                                 const uint8_t *right, size_t right_step, size_t right_frame_stride,
                                 int channels, int16_t *out, size_t out_step, size_t out_frame_stride,
                                 void *stream);

/* The same for frames that live in separate allocations: left[f] / right[f] / out[f] are the device images of frame f
 * (host arrays of n_frames device pointers, read before the call returns; one step per image kind).  This is what a
 * module adapter uses to coalesce the frames that the reference's runtime enters concurrently -- up to 12 worker threads
 * inside ImageDisparityModule::runInternal at once (cartslam.hpp:4-5, cartslam.cpp:196) -- into one launch sequence:
 * one frame per launch leaves the path-aggregation kernel latency-bound (0.67 ms per frame against 0.10 ms batched). */
int cart_compute_disparity_multi(cart_engine *engine, int n_frames,
                                 const uint8_t *const *left, size_t left_step,
                                 const uint8_t *const *right, size_t right_step, int channels,
                                 int16_t *const *out, size_t out_step, void *stream);

/* replaces: cart::disparity::interpolate (interpolation.cu:85-99) on its own; in place like the
 * reference's (the engine double-buffers internally).  min_disp16 / max_disp as disparity.hpp:27-28. */
int cart_interpolate(cart_engine *engine, int n_frames, int16_t *disp, size_t step, size_t frame_stride,
                     int radius, int iterations, int min_disp16, int max_disp, void *stream);

/* replaces: ImageDisparityDerivativeModule::runInternal (src/modules/disparity/derivative.cu:151-184):
 * out = CV_16SC2-shaped (ch0 vertical, ch1 horizontal), hist = 1x256 CV_32SC2-shaped device
 * buffer (512 int32 per frame), overwritten. */
int cart_disparity_derivative(cart_engine *engine, int n_frames,
                              const int16_t *disp, size_t disp_step, size_t disp_frame_stride,
                              int16_t *out, size_t out_step, size_t out_frame_stride,
                              int32_t *hist512, void *stream);

/* replaces: calculateDerivatives + mergeHistogram (src/modules/planeseg/planeseg.cu:31-158, launch :282-283).
 * hist256 (device) is ADDED to: with hist_frame_stride_elems == 0 all frames accumulate into one
 * persistent histogram like the module's (planeseg.hpp:160-161); with 256 each frame gets its own. */
int cart_plane_derivative_hist(cart_engine *engine, int n_frames,
                               const int16_t *disp, size_t disp_step, size_t disp_frame_stride,
                               int16_t *out, size_t out_step, size_t out_frame_stride,
                               int32_t *hist256, size_t hist_frame_stride_elems, void *stream);

/* replaces: classifyPlanes, non-temporal (planeseg.cu:160-198, launch :349-350).
 * params: one entry per frame if params_per_frame != 0, else params[0] for all. */
int cart_plane_classify(cart_engine *engine, int n_frames,
                        const int16_t *deriv, size_t deriv_step, size_t deriv_frame_stride,
                        const cart_plane_params *params, int params_per_frame,
                        uint8_t *planes, size_t planes_step, size_t planes_frame_stride, void *stream);

/* cart_plane_derivative_hist / cart_plane_classify for frames in separate allocations (host arrays of n_frames device
 * pointers, read before the call returns; one step per image kind): what a module adapter uses to serve the frames that
 * wait inside DisparityPlaneSegmentationModule::runInternal at the same moment with one launch per stage -- a one-frame
 * launch of these kernels costs 25-55 us of GPU time, a 16-frame launch 16-23 us.  hist_frame_stride_elems as above
 * (0: every frame adds to the one persistent histogram); params_per_frame: 0 = params[0] for all, 1 = params[f]. */
int cart_plane_derivative_hist_multi(cart_engine *engine, int n_frames,
                                     const int16_t *const *disp, size_t disp_step, int16_t *const *out, size_t out_step,
                                     int32_t *hist256, size_t hist_frame_stride_elems, void *stream);
int cart_plane_classify_multi(cart_engine *engine, int n_frames,
                              const int16_t *const *deriv, size_t deriv_step,
                              const cart_plane_params *params, int params_per_frame,
                              uint8_t *const *planes, size_t planes_step, void *stream);

/* replaces: the temporal-voting branch of classifyPlanes (planeseg.cu:199-240) with the tables the module builds at
 * :303-347: prev_planes[k] = unsmoothed planes of frame id-(k+1), flows[k] = optical flow of frame id-k (CV_16SC2-shaped,
 * S10.5 fixed point).  n_prev <= 8.  The pointer arrays are HOST arrays of DEVICE pointers.  Single frame: temporal
 * smoothing makes frames depend on each other, so it does not shard. */
#define CART_MAX_TEMPORAL 8
int cart_plane_temporal_vote(cart_engine *engine, const uint8_t *planes, size_t planes_step, int n_prev,
                             const uint8_t *const *prev_planes, const size_t *prev_steps,
                             const int16_t *const *flows, const size_t *flow_steps,
                             uint8_t *smoothed, size_t smoothed_step, void *stream);

/* New stage (no reference counterpart; BASELINE config 3 "plane CCL"): 4-connected components of
 * the label map over labels {0,1}; id = smallest linear index y*width+x of the component,
 * UNKNOWN pixels -> -1.  n_components (device, one int32 per frame) may be NULL. */
int cart_plane_ccl(cart_engine *engine, int n_frames,
                   const uint8_t *planes, size_t planes_step, size_t planes_frame_stride,
                   int32_t *ids, size_t ids_step, size_t ids_frame_stride,
                   int32_t *n_components, void *stream);

/* Component table of a label map and its ids (same stage, SURVEY 8a-11 "per-component {label, area, bbox}"): one entry per
 * component in ascending id order, bounding box inclusive.  `table` = device [n_frames][max_components]; a frame with more
 * components gets its first max_components entries, n_components (device, may be NULL) always holds the true count. */
typedef struct cart_component {
    int32_t id;            /* smallest linear index y*width+x of the component */
    int32_t label;         /* CART_PLANE_HORIZONTAL or CART_PLANE_VERTICAL */
    int32_t area;          /* pixels */
    int32_t x0, y0, x1, y1;
} cart_component;
int cart_plane_ccl_stats(cart_engine *engine, int n_frames,
                         const uint8_t *planes, size_t planes_step, size_t planes_frame_stride,
                         const int32_t *ids, size_t ids_step, size_t ids_frame_stride,
                         cart_component *table, int max_components, int32_t *n_components, void *stream);
/* cart_plane_ccl + cart_plane_ccl_stats in one call (same ids, count and table): the pass that writes the final ids also gathers
 * the component statistics, four launches in all instead of three + two.  `ids` must be an id map made by cart_plane_ccl /
 * cart_plane_ccl_table for cart_plane_ccl_stats to describe it: ids that are not roots of their own map are ignored there. */
int cart_plane_ccl_table(cart_engine *engine, int n_frames,
                         const uint8_t *planes, size_t planes_step, size_t planes_frame_stride,
                         int32_t *ids, size_t ids_step, size_t ids_frame_stride,
                         cart_component *table, int max_components, int32_t *n_components, void *stream);

/* replaces: HistogramPeakPlaneParameterProvider::updatePlaneParameters (planeseg.cu:405-458) +
 * util::findPeaks (src/utils/peaks.cpp:12-72).  HOST function on a host histogram.  Returns 1 if
 * *inout was updated, 0 on the reference's early-outs (parameters kept), <0 on error. */
int cart_find_plane_params(const int32_t hist256[256], cart_plane_params *inout);

/* Device-side plane-parameter schedule for the batched-frame mode: the same bookkeeping as
 * DisparityPlaneSegmentationModule::updatePlaneParameters (planeseg.cu:379-403: cumulative histogram, refresh when
 * id % update_interval == 1, reset when id % (update_interval*reset_interval) == 1) + the histogram_peak provider
 * (planeseg.cu:405-458) + util::findPeaks (peaks.cpp:12-72), replayed in frame-id order by one small kernel so
 * that a batch needs no device->host round trip.  `hists` = per-frame histograms [n_frames][256] (device, e.g. from
 * cart_plane_derivative_hist with hist_frame_stride_elems = 256, all-gathered across ranks in id order);
 * `params_out` = [n_frames] cart_plane_params (device) the frames are to be classified with.
 * provider: 0 = static (params_out = initial for every frame), 1 = histogram_peak. */
typedef struct cart_plane_schedule cart_plane_schedule;
int cart_plane_schedule_create(cart_engine *engine, int provider, const cart_plane_params *initial, int update_interval,
                               int reset_interval, cart_plane_schedule **out);
void cart_plane_schedule_destroy(cart_plane_schedule *schedule);
int cart_plane_schedule_advance(cart_plane_schedule *schedule, int first_id, int n_frames, const int32_t *hists,
                                cart_plane_params *params_out, void *stream);
/* Synchronises and copies the schedule's current parameters / cumulative histogram to the host (tests). */
int cart_plane_schedule_read(cart_plane_schedule *schedule, cart_plane_params *params_host, int32_t cum_hist_host[256]);

/* cart_plane_classify with the per-frame parameters in DEVICE memory (output of cart_plane_schedule_advance);
 * params_stride = 1 -> params[frame], 0 -> params[0] for every frame. */
int cart_plane_classify_dev(cart_engine *engine, int n_frames,
                            const int16_t *deriv, size_t deriv_step, size_t deriv_frame_stride,
                            const cart_plane_params *params_dev, int params_stride,
                            uint8_t *planes, size_t planes_step, size_t planes_frame_stride, void *stream);

/* replaces: DepthModule::runInternal (src/modules/depth.cpp:9-25): disparity x16 -> float (1/16) and
 * cv::cuda::reprojectImageTo3D(Q, 3 channels).  Q = row-major 4x4 (HOST pointer, copied), out = CV_32FC3-shaped.
 * Floating point: results are within 1e-4 relative of the CPU restatement. */
int cart_reproject_depth(cart_engine *engine, int n_frames,
                         const int16_t *disp, size_t disp_step, size_t disp_frame_stride, const float Q[16],
                         float *xyz, size_t xyz_step, size_t xyz_frame_stride, void *stream);

/* ---- superpixels ------------------------------------------------------------------------------
 * replaces: contour::ContourRelaxation + contour::createBlockInitialization as driven by SuperPixelModule
 * (src/modules/superpixels.cu:19-118, src/modules/superpixels/contourrelaxation/contourrelaxation.cu:324-447,
 * initialization.cu:13-58, features/gaussian.cu, features/compactness.cu).  The object owns the persistent label
 * image (ContourRelaxation::labelImage, contourrelaxation.hpp:47) and the per-label statistics workspaces; the
 * features are the reference's three: compactness, disparity (2-channel derivative image), colour (YCrCb).  A weight
 * <= 0 leaves the feature out (contourrelaxation.hpp:55-61).  Defaults of the JSON factory: cartconfig.cpp:121-133.
 * Semantics: oracle S13/S14 (race-free Jacobi sweeps, statistics over the whole image, fixed log sequence). */
typedef struct cart_superpixel_params {
    double direct_clique_cost;              /* 0.5 */
    double diagonal_clique_cost;            /* direct / sqrt(2) */
    double compactness_weight;              /* 0.1 */
    double progressive_compactness_cost;    /* 0.0 */
    double image_weight;                    /* 1.5 */
    double disparity_weight;                /* 1.0 */
} cart_superpixel_params;
void cart_superpixel_default_params(cart_superpixel_params *p);

typedef struct cart_superpixels cart_superpixels;
/* Allocates the state for the engine's image size and runs the block initialisation (superpixels.cu:57-59):
 * label = (y / block_h) * ceil(w / block_w) + x / block_w, max_label_id = #blocks (must be < 16384). */
int cart_superpixels_create(cart_engine *engine, const cart_superpixel_params *params, int block_w, int block_h,
                            cart_superpixels **out);
void cart_superpixels_destroy(cart_superpixels *sp);
/* Re-runs the block initialisation (the reset every `reset_iterations` frames, superpixels.cu:104-112). */
int cart_superpixels_reset(cart_superpixels *sp, void *stream);
/* ContourRelaxation::setLabelImage (contourrelaxation.cu:332-335): replaces the state with the caller's CV_16UC1
 * label image; every label must be < max_label_id (checked; synchronises `stream`). */
int cart_superpixels_set_labels(cart_superpixels *sp, const uint16_t *labels, size_t labels_step, int max_label_id,
                                void *stream);
/* ContourRelaxation::relax (contourrelaxation.cu:337-447) including the YCrCb conversion the module does first
 * (superpixels.cu:75-82): `image` = the frame's reference image, 3-channel BGR (or 1-channel gray, treated as
 * B=G=R); `deriv2` = CV_16SC2 "disparity_derivative" (may be NULL iff disparity_weight <= 0).  Runs `iterations`
 * sweeps on the state and copies the resulting label image to labels_out (CV_16UC1; may be NULL).  Calls on one
 * object are serialised in call order (the reference locks a mutex, superpixels.cu:97-99). */
int cart_superpixels_relax(cart_superpixels *sp, const uint8_t *image, size_t image_step, int channels,
                           const int16_t *deriv2, size_t deriv2_step, int iterations,
                           uint16_t *labels_out, size_t labels_out_step, void *stream);
/* superpixels_max_label blackboard value (superpixels.hpp:12). */
int cart_superpixels_max_label(const cart_superpixels *sp);

/* replaces: performSuperPixelClassifications + classifyPlanes launches (sp_planeseg.cu:27-178, 327-328).
 * deriv2 = CV_16SC2 derivative image (channel 0 is classified), labels = CV_16UC1 superpixels, max_label =
 * superpixels_max_label; n_prev/prev_planes/flows = the temporal tables the module builds (sp_planeseg.cu:243-300,
 * same layout as cart_plane_temporal_vote; n_prev = 0 -> no temporal vote).  Outputs: planes_unsmoothed = per-pixel
 * class before any vote ("planes_unsmoothed"), planes = per-superpixel majority ("planes"). */
int cart_superpixel_plane_classify(cart_engine *engine, const int16_t *deriv2, size_t deriv2_step,
                                   const uint16_t *labels, size_t labels_step, int max_label,
                                   const cart_plane_params *params, int n_prev,
                                   const uint8_t *const *prev_planes, const size_t *prev_steps,
                                   const int16_t *const *flows, const size_t *flow_steps,
                                   uint8_t *planes_unsmoothed, size_t planes_unsmoothed_step,
                                   uint8_t *planes, size_t planes_step, void *stream);

/* Stand-in for ImageOpticalFlowModule's device work (src/modules/optflow.cpp:96-140: cvtColor x2 +
 * cv::cuda::NvidiaOpticalFlow_2_0::calc(current, previous), NVIDIA fixed-function hardware): dense census block
 * matching (oracle S15).  cur / prev = the reference images of frame id and id-1 (1-channel gray or 3-channel BGR),
 * flow = CV_16SC2 in S10.5 like the reference's (include/modules/optflow.hpp:16); previous position = p - (flow >> 5).
 * radius = search range in pixels (1..16), block = half window (1..3 -> 3x3, 5x5, 7x7). */
int cart_optical_flow(cart_engine *engine, const uint8_t *cur, size_t cur_step, const uint8_t *prev, size_t prev_step,
                      int channels, int radius, int block, int16_t *flow, size_t flow_step, void *stream);

/* replaces: cv::cuda::resize(src, dst, size, 0, 0, cv::INTER_LINEAR) as KITTIDataSource::getNextInternal applies it when the
 * configured image size differs from the files' (src/sources/kitti.cpp:169-172); oracle S16.  8-bit, channels = 1 | 3,
 * device pointers, steps in bytes; needs no engine (no workspace).  device_id selects the GPU. */
int cart_resize_linear(int device_id, const uint8_t *src, size_t src_step, int src_width, int src_height, int channels,
                       uint8_t *dst, size_t dst_step, int dst_width, int dst_height, void *stream);

/* Copy between two device-visible buffers (16-byte aligned; e.g. a module output in HBM -> host memory that is mapped into
 * the device's address space, hipHostMalloc / a pinned allocation) by a kernel of `workgroups` workgroups (0 = 8).  This is
 * how a caller that wants disparity / planes in HOST memory -- the reference's consumers read cv::cuda::GpuMat, i.e. device
 * memory, so this is an extension -- gets them without a full-width copy kernel taking CUs from the compute kernels that
 * run beside it.  Asynchronous on `stream`. */
int cart_copy_narrow(cart_engine *engine, void *dst, const void *src, size_t bytes, int workgroups, void *stream);

/* replaces: util::findPeaks (peaks.cpp:12-72). HOST. Arrays hold n entries; returns #peaks, sorted by persistence. */
int cart_find_peaks(const int32_t *data, int n, int *born, int *died, int *left, int *right);

/* Test/diagnostic access to the workspace of the most recent compute call of the calling thread's
 * lease (synchronises the device).  `what`: */
enum {
    CART_DBG_GRAY_L = 0, CART_DBG_GRAY_R = 1,     /* u8  [h][w]              */
    CART_DBG_CENSUS_L = 2, CART_DBG_CENSUS_R = 3, /* u32 [h][w]              */
    CART_DBG_PATH0 = 16,                          /* +r: u8 [h][w][D], r<paths; path 1 ("up") is not materialised by
                                                     batches that take the fused WTA (D=256) */
    CART_DBG_WTA_L = 32, CART_DBG_WTA_R = 33      /* u16 [h][w]              */
};
int cart_debug_read(cart_engine *engine, int frame_slot, int what, void *host_dst, size_t bytes);

/* Test access to the integer form of the uniqueness test.  The WTA kernels replace the float compare
 * (float)S * u >= (float)best (u = (100 - uniqueness_ratio) / 100.0f) by S >= T(best); this returns T for every
 * best = 0..2047 (sums of <= 8 paths of <= 255 stay below 2048), clamped to 4095 where no reachable S passes.
 * engine != NULL: computed on its GPU by the kernels' own device function; engine == NULL: the same function
 * compiled for the host.  out2048 is a HOST array. */
int cart_debug_uniq_table(cart_engine *engine, int uniqueness_ratio, uint16_t *out2048);

/* Test / diagnostic access to the layout of the cost-slab workspace: the slots are cut into groups, each group one
 * device allocation of at most 8 GiB - 64 MiB.  Any pointer may be NULL.  group_bytes: bytes of a full group (the last one may be smaller). */
int cart_debug_slab_layout(cart_engine *engine, int *group_slots, int *n_groups, size_t *slot_bytes, size_t *group_bytes);
/* Test access: synchronises and counts the non-zero words of the component-table scratch (every table call must hand it back all
 * zeros: cart_plane_ccl_table / cart_plane_ccl_stats collect exactly what they accumulated).  *nonzero = 0 also when no table call
 * has allocated the scratch yet. */
int cart_debug_ccl_scratch_nonzero(cart_engine *engine, size_t *nonzero);

/* Per-stage device time (hipEvents recorded on the caller's stream around each stage of
 * cart_compute_disparity[_batch]).  set_timing(1) enables recording and clears the record ring
 * (the last 256 recorded calls are kept); set_timing(k), k > 1, records every k-th call only (the events
 * cost ~0.02 ms of a call's stream time: a throughput measurement that wants live stage times but not
 * their cost on every step samples); set_timing(0) stops.  collect_timing() synchronises the device and returns, per stage,
 * the MEAN milliseconds per call over the recorded calls (names are static strings).  Returns the
 * number of stages written (<= cap) and the number of calls averaged in *n_calls. */
int cart_engine_set_timing(cart_engine *engine, int enabled);
int cart_engine_collect_timing(cart_engine *engine, const char **names, float *mean_ms, int cap, int *n_calls);

/* Library / build identification ("cart_engine gfx950 <n kernels>"). */
const char *cart_engine_version(void);

#ifdef __cplusplus
}
#endif
#endif
