/*
 * mc_compute.h — C ABI of libmc_compute.so: the MI355X (gfx950) replacement for the Vulkan compute
 * runtime + GLSL shaders of pjhusky/vulkan-compute-tests.
 *
 * The reference has no FFI; the boundary this library sits behind is the C++ virtual surface of
 * `VulkanComputeApp` (src/vulkanComputeApp.h:30-67) as driven by src/main.cpp:28-33.  Each entry
 * point below cites the reference code it replaces (paths relative to the reference checkout).
 * The C++ mirror of the reference interface lives in vulkan-compute-tests_amd/host/ and calls only
 * these functions; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes, no exceptions, no STL, no torch types.  Every function
 * returns MC_OK (0) or an mc_status error code; mc_error_string() renders it; mc_last_error_detail()
 * gives the HIP/RCCL message of the last failure on the calling thread.  All calls are blocking
 * unless the name ends in _async.  A context is not thread-safe; use one per thread/device.
 *
 * Buffer layout (the reference's storage-buffer contract): row-major, one `vec4` fp32 per pixel
 * (16 B, src/mandelbrotApp.h:187-189, shaders/mandelbrot.comp:10-17,59, shaders/pathTracer.comp:71),
 * rows are STORAGE rows (for the path tracer storage row r holds pix.y = H-1-r, pathTracer.comp:349).
 * Tile calls take [row_begin,row_end) in storage rows and a pointer to the FIRST ROW OF THE TILE.
 */
#ifndef MC_COMPUTE_H_
#define MC_COMPUTE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (end of round 6): + mc_mandelbrot_render_banded, row bands in mc_mandelbrot_render_rgba8, scene-class bit 32 (MC_PT_SCENE_SPECULAR),
 * bit 1 of mc_context_warmup_mandelbrot's last argument.
 * 2 (round 6): + mc_assemble_rgba8_device_async, mc_context_warmup_*; since 1 (round 5 additions, un-bumped then): mc_build_id,
 * mc_host_alloc / mc_host_free, mc_context_last_timing, math_mode 2, scene-class bit 16; the measurement flag enums moved to
 * mc_compute_test.h.  A binder checks mc_abi_version() against the MC_ABI_VERSION it was written for. */
#define MC_ABI_VERSION 3

typedef enum mc_status {
    MC_OK = 0,
    MC_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, zero size, row range outside the image ...          */
    MC_ERR_NO_DEVICE = 2,        /* no HIP device / device index out of range (vulkanComputeApp.cpp:78) */
    MC_ERR_HIP = 3,              /* a HIP runtime call failed; see mc_last_error_detail()             */
    MC_ERR_RCCL = 4,             /* an RCCL call failed                                               */
    MC_ERR_UNSUPPORTED = 5,      /* e.g. scene larger than the on-chip scene store                    */
    MC_ERR_OUT_OF_MEMORY = 6     /* a device allocation failed; mc_host_alloc: more than this process may still take */
} mc_status;

typedef struct mc_context mc_context; /* opaque: device, stream, scratch, LUT cache, (optional) RCCL comms */

/* ---- lifecycle: replaces VulkanComputeApp::init() (vulkanComputeApp.cpp:443-449: createInstance,
 *      findPhysicalDevice, createDevice) and cleanupVulkanResources() (:673-695) -------------------- */
int mc_abi_version(void);
/* Which build is this: "pt=<id> mandel=<id> lib=<id>", each id 16 hex digits of the SHA-256 of the sources that kernel family (path
 * tracer, Mandelbrot, whole library) was compiled from, and of the compiler flags.  Measurement records kept beside the code
 * (profiles/ *_pmc_summary.json) carry the id of the library they were taken on; bench.py quotes them only for a matching build. */
const char* mc_build_id(void);
int mc_device_count(int* count);
int mc_context_create(int device, mc_context** out_ctx);
int mc_context_destroy(mc_context* ctx);
const char* mc_error_string(int status);
const char* mc_last_error_detail(void);
/* Device name / CU count of the context's device (vulkanComputeApp.cpp:163 picks devices[0]). */
int mc_context_device_info(mc_context* ctx, char* name, size_t name_len, int* compute_units, int* clock_khz);

/* Page-locked host memory for the storage buffer the application owns — what stands where the reference allocates its output buffer
 * HOST_VISIBLE | HOST_COHERENT (VulkanComputeApp::createBuffer, vulkanComputeApp.cpp:489-533; mapped by getRenderedImage,
 * mandelbrotApp.h:153 / pathtracerApp.h:206; freed at vulkanComputeApp.cpp:684-685).  The buffer lives in HBM while the kernels
 * write it, so 16 B/pixel cross PCIe once, at the end of mc_*_render; into page-locked memory that copy needs no staging and no
 * page pinning by the runtime.  Buffers of 32 MB and more are huge-page mappings first-touched in parallel and then registered with
 * the runtime (K4's 629 MB: 4 ms instead of hipHostMalloc's 86, same copy rate; profiles/r06_hostmem_probe.txt), smaller ones
 * hipHostMalloc.  Any host pointer is accepted by the render calls — a pageable one is copied at nearly the same rate once its pages
 * are resident, at half of it while they are not — so these two are an ownership convention, not a requirement.
 * Usable before any context exists; the memory is visible to every device of the node.  Every page handed out is touched or pinned:
 * a request beyond what the process may still take (MemAvailable, the room under a cgroup memory limit) returns MC_ERR_OUT_OF_MEMORY
 * before a page is touched (mc_last_error_detail names both figures) — it would otherwise meet the out-of-memory killer. */
int mc_host_alloc(size_t bytes, void** out_ptr);
int mc_host_free(void* ptr);

/* Device time of the LAST blocking host-buffer call on this context (mc_mandelbrot_render, mc_pathtrace_render, mc_*_render_rgba8):
 * kernel_ms = first launch to last kernel end (render, and the on-device conversion of the _rgba8 forms), copy_ms = the device -> host
 * copy that follows.  HIP events on the context's stream; either pointer may be NULL.  MC_ERR_INVALID_ARGUMENT before the first such
 * call.  (The reference times nothing, vulkanComputeApp.cpp:451-466 only prints progress; the apps print these next to run().) */
int mc_context_last_timing(mc_context* ctx, double* kernel_ms, double* copy_ms);

/* Shader clock (MHz) the device holds with every SIMD busy on fp32 VALU work, measured in-kernel (s_memtime against the
 * constant 100 MHz s_memrealtime over ~2 ms).  MI355X boxes differ by >10 % here (DVFS), and VALU-issue-bound kernels with
 * them; bench.py prints it next to every timing so that runs on different boxes can be compared (no reference counterpart). */
int mc_context_measure_clock(mc_context* ctx, double* sclk_mhz);

/* ---- Mandelbrot: replaces shaders/mandelbrot.comp:21-60 + the dispatch recorded in
 *      MandelbrotApp::createCommandBuffer (src/mandelbrotApp.h:137-147) ----------------------------- */
enum { MC_PRECISION_F32 = 0, MC_PRECISION_DS = 1 /* two-float, emulateDouble.h.glsl:59-139 */ };
enum {
    /* bit 0 is a measurement switch of this repository (include/mc_compute_test.h), never set by a binding */
    MC_MANDEL_ITERS_U16 = 1u << 1 /* device form: d_iters is a uint16_t plane (max_iter <= 65535) — the multi-GPU exchange  */
                                  /* format, half of the 4-B plane and an eighth of the vec4 (mc_mandelbrot_assemble_...)  */
};

typedef struct mc_mandelbrot_params {
    uint32_t width, height;   /* WIDTH/HEIGHT (mandelbrot.comp:5-6); reference 2000x2000 (main.cpp:20)  */
    uint32_t max_iter;        /* M (mandelbrot.comp:40); reference 128                                   */
    uint32_t precision;       /* MC_PRECISION_*                                                         */
    /* c = centre + (uv - 0.5) * scale (mandelbrot.comp:38); reference centre (-0.445, 0), scale 2.34   */
    /* on both axes.  *_lo are the low words for MC_PRECISION_DS (hi=(float)d, lo=(float)(d-hi)).       */
    float centre_x_hi, centre_x_lo, centre_y_hi, centre_y_lo;
    float scale_x_hi, scale_x_lo, scale_y_hi, scale_y_lo;
    float k_color[4];         /* push constant kColor (mandelbrotApp.h:139); reference {0.1,0.7,0.6,0}   */
    uint32_t row_begin, row_end; /* storage rows rendered by this call; 0,height for the whole image     */
    uint32_t row_block, row_stride; /* 0,0: the tile is the contiguous rows [row_begin,row_end).          */
                              /* Otherwise the tile is the rows row_begin + k*row_stride + j (j<row_block,   */
                              /* < row_end), stored compactly: tile row k*row_block + j (interleaved row     */
                              /* blocks, one residue class per GPU; see mc_tile_rows)                        */
    uint32_t flags;           /* MC_MANDEL_*                                                            */
    uint32_t reserved;
} mc_mandelbrot_params;

/* Fills p with the reference defaults for a W x H image (main.cpp:20, mandelbrot.comp:38-40). */
int mc_mandelbrot_default_params(uint32_t width, uint32_t height, mc_mandelbrot_params* p);

/* Host-buffer form (what MandelbrotApp::run + vkMapMemory give the reference, mandelbrotApp.h:149-155):
 * out_rgba_f32: (row_end-row_begin)*width*4 floats, may be NULL; out_iters: same pixel count of
 * uint32 iteration counts n in [0,max_iter], may be NULL.  At least one must be non-NULL. */
int mc_mandelbrot_render(mc_context* ctx, const mc_mandelbrot_params* p, float* out_rgba_f32, uint32_t* out_iters);

/* Device-buffer form (buffers already resident in HBM; pointers are device pointers on ctx's device).
 * Asynchronous on `stream` (a hipStream_t, NULL = the context's stream); either output may be NULL. */
int mc_mandelbrot_render_device_async(mc_context* ctx, const mc_mandelbrot_params* p, void* d_rgba_f32,
                                      void* d_iters, void* stream);

/* The (max_iter+1)-entry colour table: entry n = vec4 written for iteration count n
 * (mandelbrot.comp:50-56, evaluated on the host in fp32 source order).  lut_f32: (max_iter+1)*4. */
int mc_mandelbrot_colour_lut(uint32_t max_iter, const float k_color[4], float* lut_f32);

/* ---- Path tracer: replaces shaders/pathTracer.comp:343-458 and the spp-dispatch loop of
 *      PathtracerApp::createCommandBuffer (src/pathtracerApp.h:358-378), fused into one launch ------ */
enum {
    MC_PT_MATH_STRICT = 0, /* IEEE div/sqrt + the explicit "mc math" sin/cos/pow: bit-identical to the oracle */
    MC_PT_MATH_FAST = 1,   /* toleranced parity (RMSE <= 0.5, 99.9-percentile per-pixel L2 <= 4 of 255 at 500 spp).  */
                           /* The library renders the request with the tier MEASURED to hold that bound on scenes like the one given: */
                           /* the fast tier (gfx950 hardware rcp/rsq/sqrt/sin/cos/exp/log, a*b+c contracted) for a scene with no more  */
                           /* specular surface than the reference scene's (mc_pathtrace_scene_class: up to three spheres, diffuse     */
                           /* walls, mirror / glass spheres no larger) — the reference scene reads 2.49, 174 random scenes of that     */
                           /* class at most 3.8 but for two at 4.2: met where it was stated, measured and not guaranteed around it;     */
                           /* the careful tier below everywhere else (at most 1.8 on every scene measured); the strict kernels for a  */
                           /* light all but enclosed by an opaque sphere.  mc_pathtrace_select_kernel reports which                     */
                           /* (mc_pathtrace_kernel_info.math_mode).  A caller that needs the margin on EVERY scene asks for the tier below. */
    MC_PT_MATH_FAST_CAREFUL = 2 /* the fast mode's careful tier on request: the same kernels and shortcuts, division / sqrt / 1/sqrt */
                           /* rounded as the reference rounds them and no contraction — a sample differs from the reference's by far  */
                           /* fewer roundings and takes another path correspondingly less often; 1.14 - 1.43 x the fast tier's time  */
};

/* mc_pathtrace_params.flags: MC_PT_PRECISION(x) below (bits 16-19) is the one field a binding sets.  Bits 0-15 belong to this
 * repository's measurement tools (kernel-selection A/B switches, include/mc_compute_test.h: not part of the boundary — a binding
 * leaves them zero, and then every request is rendered by the kernel the host selects, inside the mode's parity contract);
 * every other bit is reserved and refused with MC_ERR_INVALID_ARGUMENT. */
/* Sphere-test precision branch of pathTracer.comp:132-256.  The reference compiles every variant OUT
 * (emulateDouble.h.glsl:13-26 are all FALSE) and enables one by hand together with the
 * TEST_PRECISION_WITH_LARGE_SPHERE_WALLS scene (pathtracerApp.h:11,28-35).  This DOES change results. */
enum {
    MC_PT_PREC_F32 = 0,  /* the default build: fp32 test only (pathTracer.comp:316-331)              */
    MC_PT_PREC_FP64 = 1, /* USE_NATIVE_FP64 (pathTracer.comp:132-143)                                 */
    MC_PT_PREC_DS = 2,   /* DS_f32_f32 (pathTracer.comp:144-213, emulateDouble.h.glsl:28-223)          */
    MC_PT_PREC_DF64 = 3  /* DF64_F32_F32 (pathTracer.comp:214-256, emulateDouble.h.glsl:225-356)       */
};
#define MC_PT_PRECISION(x) ((uint32_t)(x) << 16)

typedef struct mc_pathtrace_params {
    uint32_t width, height;            /* push constant imgdim (pathtracerApp.h:44-47,58-59)            */
    uint32_t spp;                      /* push constant samps.y (pathtracerApp.h:61; main.cpp:22)        */
    uint32_t sample_begin, sample_end; /* samps.x range rendered by this call; 0,spp = whole render.     */
                                       /* sample_begin>0 continues the accumulator already in the buffer */
    uint32_t max_depth;                /* maxDepth, 12 (pathTracer.comp:367)                             */
    uint32_t row_begin, row_end;       /* storage rows; 0,height for the whole image                    */
    uint32_t row_block, row_stride;    /* interleaved row blocks, as in mc_mandelbrot_params             */
    uint32_t math_mode;                /* MC_PT_MATH_*                                                   */
    uint32_t flags;
} mc_pathtrace_params;

int mc_pathtrace_default_params(uint32_t width, uint32_t height, uint32_t spp, mc_pathtrace_params* p);

/* The reference's default scene tables (pathtracerApp.h:14-39): 12 floats per object
 * (plane: equation.xyzw | emission.xyz0 | colour.rgb,material; sphere: centre.xyz,radius | ... ). */
int mc_pathtrace_default_scene(const float** planes, uint32_t* n_planes, const float** spheres, uint32_t* n_spheres);

/* Host-buffer form: planes/spheres are the host tables PathtracerApp::preRun memcpy's into its SSBOs
 * (pathtracerApp.h:152-161,189-198); out_rgba_f32 receives (row_end-row_begin)*width*4 floats. */
int mc_pathtrace_render(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                        const float* spheres, uint32_t n_spheres, float* out_rgba_f32);

/* Device-buffer form: d_rgba_f32 is a device pointer to the tile; scene tables are still host
 * pointers (432 B, passed as kernel constants).  Asynchronous on `stream` (NULL = context stream). */
int mc_pathtrace_render_device_async(mc_context* ctx, const mc_pathtrace_params* p, const float* planes,
                                     uint32_t n_planes, const float* spheres, uint32_t n_spheres, void* d_rgba_f32,
                                     void* stream);

/* ---- Host post-process on the GPU: replaces the scalar loops of
 *      getRenderedImage (mandelbrotApp.h:149-170, pathtracerApp.h:202-223) and the 180-degree
 *      rotation (pathtracerApp.h:236-243).  u8 = (uint8_t)(scale*c) with the x86-64 semantics the
 *      reference binary has (cvttss2si, low byte), alpha = 255.  rotate180 != 0 applies the PT swap. */
int mc_convert_rgba8_device_async(mc_context* ctx, const void* d_rgba_f32, uint32_t width, uint32_t height,
                                  float scale, int rotate180, void* d_rgba8, void* stream);
int mc_convert_rgba8(mc_context* ctx, const float* rgba_f32, uint32_t width, uint32_t height, float scale,
                     int rotate180, uint8_t* rgba8);
/* Render + post-process fused on the device: the whole image is rendered, converted exactly as the reference's
 * saveRenderedImage would (Mandelbrot: scale 255, mandelbrotApp.h:159-174; path tracer: scale 1 and the
 * 180-degree rotation, pathtracerApp.h:202-243) and only the RGBA8 image (4 B/pixel instead of 16) is copied to
 * out_rgba8 (width*height*4 bytes, host).  The path tracer: whole image only (row_begin = 0, row_end = height, no interleave).
 * The Mandelbrot set, which is not rotated: also a contiguous band of rows [row_begin, row_end) (no interleave) — out_rgba8 then receives
 * (row_end-row_begin)*width*4 bytes, the band's rows of the whole image's RGBA8 (the app renders band by band while its PNG workers run). */
int mc_mandelbrot_render_rgba8(mc_context* ctx, const mc_mandelbrot_params* p, uint8_t* out_rgba8);
/* The same image (rows [row_begin, row_end), no interleave) rendered in bands of band_rows rows, PIPELINED: band k + 1 is launched on a second
 * stream before band k has finished — it fills the device while band k's last tiles drain and while band k's rows travel to the host — and
 * on_rows(rows_done, user) is called on the calling thread as each band has ARRIVED, in order (rows [row_begin, rows_done) of the output are
 * final): the caller's own work on the image — the app's PNG workers — runs beside the rest of the render.  Exactly one of out_rgba_f32
 * (the storage buffer, 16 B/pixel) and out_rgba8 (converted on the device, 4 B/pixel) is non-NULL; on_rows may be NULL.  Blocking; the
 * bytes are those of mc_mandelbrot_render / mc_mandelbrot_render_rgba8.  mc_context_last_timing then reports kernel = first launch to the end
 * of the last kernel, copy = what of the copies was not hidden behind a kernel.  mc_context_warmup_mandelbrot(ctx, p, rgba8 | 2) prepares it. */
typedef void (*mc_rows_ready_fn)(uint32_t rows_done, void* user);
int mc_mandelbrot_render_banded(mc_context* ctx, const mc_mandelbrot_params* p, float* out_rgba_f32, uint8_t* out_rgba8, uint32_t band_rows,
                                mc_rows_ready_fn on_rows, void* user);
int mc_pathtrace_render_rgba8(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                              const float* spheres, uint32_t n_spheres, uint8_t* out_rgba8);

/* ---- cold start (no reference counterpart: vkCreateComputePipelines compiles the shader inside preRun, vulkanComputeApp.cpp:589-643,
 *      before anything is timed; here the runtime loads a kernel family's code object on its first launch, 8 - 12 ms, and the first
 *      full-size launch would pay for it) -----------------------------------------------------------------------------------------
 * mc_context_warmup_*: everything the request's first launch would otherwise do once — device scratch for the whole request (and the
 * RGBA8 image when rgba8 != 0), timing events, scene / colour / c tables, and a 16 x 8 (one-tile) launch of the kernel family the
 * request selects, so that the code object is resident.  Blocks the CALLING thread for the load and returns with the tiny launch
 * queued on the context's stream; results are unaffected (the scratch it touches is overwritten by the render).  An application calls
 * it from a helper thread while it does other start-up work (allocating its storage buffer, opening its output), and joins that
 * thread before its next call on the context (a context is not thread-safe).  (Round 6 also tried to move the storage buffer's
 * allocation BEHIND the launch with a two-phase render call: registering host memory while a kernel runs stalls the device — K4's
 * kernel 82 ms instead of 60 — so the buffer is made first, in 4 ms, and the blocking calls stayed as they were.)
 * mc_context_warmup_mandelbrot's `rgba8`: bit 0 as above, bit 1 = mc_mandelbrot_render_banded will follow (its second stream is made now). */
int mc_context_warmup_pathtrace(mc_context* ctx, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                                const float* spheres, uint32_t n_spheres, int rgba8);
int mc_context_warmup_mandelbrot(mc_context* ctx, const mc_mandelbrot_params* p, int rgba8);

/* Host-side analysis the path tracer applies to a scene before choosing a kernel; touches no device, usable without a
 * GPU.  *out_class: bit 0 (MC_PT_SCENE_SLAB) — six axis-aligned planes in index order x,x,y,y,z,z plus one to eight spheres
 * (the reference scene, pathtracerApp.h:14-39, has three): the specialised slab kernels run; bit 1 (MC_PT_SCENE_LIGHTS_INSIDE) —
 * additionally the planes close a box, the camera (pathTracer.comp:352) and every emissive sphere lie inside it with a
 * margin: shadow rays (pathTracer.comp:420) skip the plane tests.  Both specialisations are bit-exact;
 * every other scene takes the generic kernel.  Bit 2 (MC_PT_SCENE_SPHERES_DISJOINT) — slab scenes: the spheres are pairwise
 * disjoint with a margin; the fast sample-pool kernel then decides shadow rays without square roots, ordering the spheres a ray meets
 * by the projections of their centres (overlapping spheres: its root form; strict math does not depend on it).  Bit 3 (MC_PT_SCENE_LIGHT_ENCLOSED) — any scene: an
 * emissive sphere intersects a non-emissive diffuse sphere (or comes within 1.5 of its own radii of it), or is all but enclosed by a
 * mirror sphere.  Next-event estimation at point-blank range through rays grazing the sphere they start on (pathTracer.comp:325-327,
 * 420) makes such an image a collection of near-ties, which fast math decides differently from the reference arithmetic far more
 * often than its tolerance allows: an MC_PT_MATH_FAST request for such a scene is RENDERED WITH THE STRICT KERNELS
 * (bit-identical to the oracle), never silently outside the bound.  Bit 4 (MC_PT_SCENE_MANY_SPHERES) — any scene: four or more spheres.
 * The share of fast-math samples that take another path than the reference's grows with the number of (specular) spheres a path can
 * run through; on random boxes the fast tier holds the bound with three spheres (36 scenes, at most 3.3 of 4.0), misses it on 2 of 44 with
 * four (5.5, 6.3), reads 3.2 with five and 4.4 .. 5.6 from six on (profiles/r05_fork_census_careful.txt, r06_fast_tier_{3,4}_spheres.txt): an
 * MC_PT_MATH_FAST request for such a scene is rendered by the careful tier (MC_PT_MATH_FAST_CAREFUL).  Bit 5 (MC_PT_SCENE_SPECULAR) — any
 * scene with up to three spheres that has more specular surface than the reference scene (pathtracerApp.h:14-39: diffuse walls, one mirror
 * and one glass sphere of r = 0.8): a mirror or glass wall, or mirror spheres (material 2), or glass spheres (material 3), whose squared radii
 * sum to more than 0.65.  A forked sample that a specular chain carries to a light moves its pixel by the light's whole emission: of 276
 * jittered three-sphere rooms eighteen are outside the bound in the fast tier (up to 8.0) — fourteen of the 132 with a specular wall, two of
 * the 38 with larger mirror spheres, TWO of the 106 this bit leaves to the fast tier (4.2; the others at most 3.8; 76 of them drawn after
 * the rule was set: profiles/r06_fast_tolerance_scenes*.txt).  The careful tier renders the 170 others at 0.8 or less. */
#define MC_PT_SCENE_SLAB 1u
#define MC_PT_SCENE_LIGHTS_INSIDE 2u
#define MC_PT_SCENE_SPHERES_DISJOINT 4u   /* slab scenes: the spheres are pairwise disjoint (the fast pool kernel then needs no square roots for shadow rays) */
#define MC_PT_SCENE_LIGHT_ENCLOSED 8u     /* a light intersecting a diffuse sphere / all but enclosed by a mirror: fast math requests are rendered strict */
#define MC_PT_SCENE_MANY_SPHERES 16u      /* any scene with four or more spheres: an MC_PT_MATH_FAST request is rendered by the careful tier            */
#define MC_PT_SCENE_SPECULAR 32u          /* up to three spheres, more specular surface than the reference scene's: likewise the careful tier          */
int mc_pathtrace_scene_class(const float* planes, uint32_t n_planes, const float* spheres, uint32_t n_spheres,
                             uint32_t* out_class);

/* Which kernel mc_pathtrace_render* will run for a request — decided on the host from the parameters and the scene alone (no
 * device, usable without a GPU).  In MC_PT_MATH_FAST different kernels are different instruction sequences (equal within the
 * tolerance, not bit for bit), so an N-GPU or progressive caller that needs tiles / ranges to compose bit-identically asserts
 * that every part reports the same `kernel` and `lanes_per_pixel` as the whole image does.  Validates like the render call. */
enum {
    MC_PT_KERNEL_GENERIC = 0,        /* any scene, records staged into LDS (pathTracer.comp:112-131 as written)              */
    MC_PT_KERNEL_SLAB = 1,           /* 6 axis-aligned planes + 3 spheres, round-synchronous                                 */
    MC_PT_KERNEL_BOX = 3,            /* closed-box scene facts at compile time, round-synchronous (fast math)                */
    MC_PT_KERNEL_POOL = 4,           /* the sample-pool kernel (the default for the reference scene, both math modes)        */
    MC_PT_KERNEL_GENERIC_MEMORY = 5  /* any scene, records read from memory (large scenes)                                   */
};
typedef struct mc_pathtrace_kernel_info {
    uint32_t kernel;          /* MC_PT_KERNEL_*                                                                             */
    uint32_t lanes_per_pixel; /* sample-parallel width S: lanes of a wave that share a pixel (1, 4 or 16)                    */
    uint32_t math_mode;       /* the mode that RUNS: for a fast request MC_PT_MATH_FAST, MC_PT_MATH_FAST_CAREFUL (MC_PT_SCENE_MANY_SPHERES,  */
                              /* MC_PT_SCENE_SPECULAR) or MC_PT_MATH_STRICT (MC_PT_SCENE_LIGHT_ENCLOSED)                                    */
    uint32_t launches;        /* kernel launches per call (2: a ragged sample count in the round-synchronous kernels)        */
} mc_pathtrace_kernel_info;
int mc_pathtrace_select_kernel(const mc_pathtrace_params* p, const float* planes, uint32_t n_planes, const float* spheres,
                               uint32_t n_spheres, mc_pathtrace_kernel_info* out);

/* ---- stream / tiling helpers ------------------------------------------------------------------ */
int mc_context_synchronize(mc_context* ctx);
/* Rows per interleave block used by every multi-GPU path of this library (mc_multi_*, bench.py's sharding): rank r of n
 * owns the storage rows whose block index (row / mc_row_block()) is congruent to r mod n.  */
uint32_t mc_row_block(void);
/* Number of storage rows in the tile described by (row_begin,row_end,row_block,row_stride). */
uint32_t mc_tile_rows(uint32_t row_begin, uint32_t row_end, uint32_t row_block, uint32_t row_stride);
/* Reassembles the storage buffer from n_tiles interleaved tiles laid out back to back, each padded to
 * tile_rows_padded rows (the layout an RCCL gather of equal-sized tiles leaves on the root): tile t holds
 * the rows t*row_block + k*n_tiles*row_block + j.  bytes_per_pixel is 16 (vec4 fp32) or 4 (iteration counts). */
int mc_deinterleave_rows_device_async(mc_context* ctx, const void* d_tiles, uint32_t width, uint32_t height,
                                      uint32_t n_tiles, uint32_t row_block, uint32_t tile_rows_padded,
                                      uint32_t bytes_per_pixel, void* d_out, void* stream);

/* Path-tracer RGBA8 exchange on the root (4 B/pixel cross xGMI): d_tiles_rgba8 holds n_tiles interleaved tiles of
 * RGBA8 pixels, each converted by its owner with mc_convert_rgba8_device_async(rotate180 = 0), laid out as
 * mc_deinterleave_rows_device_async expects; writes the whole image d_rgba8 (width*height*4 bytes) in storage-row order and, with
 * rotate180 != 0, point-reflected as PathtracerApp::saveRenderedImage leaves it (pathtracerApp.h:236-243, incl. an odd width's
 * untouched middle column) — the same bytes as mc_pathtrace_render_rgba8 of the whole image on one GPU. */
int mc_assemble_rgba8_device_async(mc_context* ctx, const void* d_tiles_rgba8, uint32_t width, uint32_t height, uint32_t n_tiles,
                                   uint32_t row_block, uint32_t tile_rows_padded, int rotate180, void* d_rgba8, void* stream);

/* Mandelbrot exchange on the root: d_tiles holds n_tiles interleaved tiles of ITERATION COUNTS (iters_bytes = 2: uint16_t,
 * MC_MANDEL_ITERS_U16; 4: uint32_t) laid out as mc_deinterleave_rows_device_async expects; writes the whole image's storage
 * buffer d_rgba_f32 (lut[n] per pixel: exactly what the render kernel writes for that count, mandelbrot.comp:50-59) and/or its
 * uint32 count plane d_iters.  The ranks of a multi-GPU render then exchange 2-4 B/pixel instead of 16. */
int mc_mandelbrot_assemble_device_async(mc_context* ctx, const mc_mandelbrot_params* p, const void* d_tiles, uint32_t iters_bytes,
                                        uint32_t n_tiles, uint32_t row_block, uint32_t tile_rows_padded, void* d_rgba_f32,
                                        void* d_iters, void* stream);

/* ---- single-process multi-GPU render (north_star: row tiles + RCCL gather to rank 0) --------------
 * Renders the whole image on n_devices GPUs of this node (interleaved row blocks), gathers
 * the fp32 tiles to device 0 with RCCL over xGMI and copies the assembled storage buffer to
 * out_rgba_f32 (host, width*height*4 floats).  n_devices = 1 degenerates to the single-GPU path. */
typedef struct mc_multi mc_multi;
int mc_multi_create(int n_devices, mc_multi** out);
int mc_multi_destroy(mc_multi* m);
int mc_multi_mandelbrot_render(mc_multi* m, const mc_mandelbrot_params* p, float* out_rgba_f32, uint32_t* out_iters);
int mc_multi_pathtrace_render(mc_multi* m, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                              const float* spheres, uint32_t n_spheres, float* out_rgba_f32);
/* As mc_*_render_rgba8: only RGBA8 leaves the GPU.  Mandelbrot: device 0 converts the storage buffer it rebuilt from the gathered
 * counts.  Path tracer: every device converts its own tile, the gather moves 4 B/pixel, device 0 de-interleaves and applies the
 * point reflection on bytes (mc_assemble_rgba8_device_async). */
int mc_multi_mandelbrot_render_rgba8(mc_multi* m, const mc_mandelbrot_params* p, uint8_t* out_rgba8);
int mc_multi_pathtrace_render_rgba8(mc_multi* m, const mc_pathtrace_params* p, const float* planes, uint32_t n_planes,
                                    const float* spheres, uint32_t n_spheres, uint8_t* out_rgba8);

#ifdef __cplusplus
}
#endif
#endif /* MC_COMPUTE_H_ */
