/*
 * mfx.h -- C ABI of the MI355X-native per-voxel fingerprint matcher.
 *
 * The reference (rensonnetg/microstructure_fingerprinting) is pure Python and has no
 * FFI layer: the drop-in boundary is its Python call signatures.  Each entry point
 * below is what a reference maintainer would bind (ctypes stub in INTEGRATION.md)
 * behind one of those signatures; the reference interface it replaces is cited as
 * file:line relative to /root/reference/microstructure_fingerprinting/.
 *
 * Conventions
 *   - plain pointers and sizes only; all matrices row-major, float64; indices int32/int64.
 *   - every function returns 0 on success or an MFX_ERR_* code; mfx_last_error() gives
 *     the message (thread-local).  The Python wrapper maps codes to the exception
 *     classes the reference raises (AssertionError / ValueError / RuntimeError).
 *   - the caller owns every input/output buffer; the library copies host inputs to the
 *     device and never keeps a host pointer after return.  Opaque handles (mfx_tables,
 *     mfx_plan) are library-owned and freed by their *_destroy function.
 *   - "_dev" variants take DEVICE pointers (HBM-resident inputs/outputs, e.g. torch
 *     tensors' data_ptr) and a hipStream_t passed as void*; they only enqueue kernels on that
 *     stream and do not wait for the device.  (Scratch memory comes from an arena the calling
 *     thread keeps per stream; only while that arena still grows - the first call or two of a
 *     given size - does a call allocate device memory, which synchronises.)  What the
 *     reference would raise from inside its voxel loop (a fascicle direction that is not a
 *     unit vector, mf_utils.py:1798-1802) cannot be returned by an asynchronous call: the
 *     kernels flag it in the plan's status word, mfx_plan_status() reports it.  Exception:
 *     mfx_monte_carlo_average_dev returns its (small) result in a host array and therefore
 *     waits for its own work.
 *   - all library state that is not owned by a handle (last error, timing events, diagnostic
 *     switches, hand-back counters) is per host thread: one host thread drives one GPU.
 *   - there is no CPU fallback: every compute entry point fails with MFX_ERR_NO_DEVICE
 *     when no gfx950 device is usable.
 */
#ifndef MFX_H
#define MFX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MFX_OK 0
#define MFX_ERR_ARG 1          /* malformed argument (maps to AssertionError / ValueError) */
#define MFX_ERR_G_RANGE 2      /* mf_utils.py:1829-1836 "Extrapolation not supported" -> ValueError */
#define MFX_ERR_NO_DEVICE 3    /* no usable HIP device -> RuntimeError */
#define MFX_ERR_HIP 4          /* HIP runtime failure -> RuntimeError */
#define MFX_ERR_UNSUPPORTED 5  /* shape outside a kernel's limits -> NotImplementedError.  The voxel loop (mfx_fit_batch*) does
                                * not return it for large dictionaries, long protocols or many CSF+EAR columns: those classes
                                * run voxel by voxel through the explicit-dictionary solver (slow, same results); it remains for
                                * mfx_solve_exhaustive (> 8 sub-dictionaries, > 2^40 index tuples, > 30000 columns) and the
                                * launch limits of mfx_monte_carlo_average */
#define MFX_ERR_DIR_NORM 6     /* mf_utils.py:1798-1802 non-unit fascicle direction -> ValueError */

typedef struct mfx_tables mfx_tables; /* device-resident per-shell knot tables */
typedef struct mfx_plan mfx_plan;     /* device-resident per-protocol row plan   */

const char* mfx_last_error(void);
int mfx_device_count(void);
/* library/ABI version, bumped on any signature change or new entry point (3: mfx_fit_batch_volume, mfx_volume_rows, counters 8-10 of
 * mfx_debug_last_counter, mfx_debug_set_k3_cap, mfx_debug_set_force_generic, mfx_debug_set_k3_screen) */
int mfx_abi_version(void);

/* ---- tables: replaces the interpolator objects returned by
 * init_PGSE_multishell_interp (mf_utils.py:1959-2085; dict keys at 2081-2085).
 * knots_x   [P]      ascending within each shell (P = shell_off[S])
 * knots_Y   [P x N]  row-major dictionary rows at the knots
 * shell_off [S+1]
 * G_un      [S]      sorted unique gradient strengths of the dense scheme
 * The per-interval slopes (Y[j]-Y[j-1])/(x[j]-x[j-1]) of SciPy's interp1d._call_linear
 * are precomputed once here (they are direction independent).                     */
int mfx_tables_create(const double* knots_x, const int32_t* shell_off, const double* knots_Y,
                      const double* G_un, int S, int N, int device, mfx_tables** out);
void mfx_tables_destroy(mfx_tables* t);
int mfx_tables_num_atoms(const mfx_tables* t);

/* ---- plan: per-protocol row -> shell mapping.
 * multishell: mf_utils.py:1810-1839 (exact float equality of G, else bracketing pair,
 * MFX_ERR_G_RANGE outside the table range).  scheme [M x 7] = [gx gy gz G Delta delta TE].
 * explicit: rotate_atom (mf_utils.py:1205-1437): the caller has already grouped rows
 * into (G,Delta,delta) shells; gdirs [M x 3] are used as given, shell_of_row [M].  With such a
 * plan every fascicle direction is divided by its norm before use, in mfx_rotate* and in the
 * voxel loop alike (rotate_atom normalises newdir, mf_utils.py:1262-1270, and has no unit-norm check). */
int mfx_plan_create_multishell(const mfx_tables* t, const double* scheme, int M, mfx_plan** out);
int mfx_plan_create_explicit(const mfx_tables* t, const double* gdirs, const int32_t* shell_of_row, int M,
                             mfx_plan** out);
void mfx_plan_destroy(mfx_plan* p);
/* Deferred error channel of the asynchronous entry points: waits for `stream`, reports and clears what the fit
 * kernels launched with this plan have flagged since the last call: MFX_ERR_DIR_NORM if a voxel's fascicle
 * direction failed the reference's unit-norm check (mf_utils.py:1798-1802, raised there per voxel), else MFX_OK. */
int mfx_plan_status(const mfx_plan* p, void* stream);

/* ---- batched rotation: B directions -> out [B x M x N].
 * Replaces interp_PGSE_from_multishell(sch_mat, newdir, msinterp=...) (mf_utils.py:1693-1956)
 * and the evaluation step of rotate_atom (mf_utils.py:1423-1426).
 * normalise_dirs != 0 divides each direction by its norm first (rotate_atom, mf_utils.py:1269). */
int mfx_rotate(const mfx_plan* p, const double* dirs, int64_t B, int normalise_dirs, double* out);
int mfx_rotate_dev(const mfx_plan* p, const double* d_dirs, int64_t B, int normalise_dirs, double* d_out,
                   void* stream);
/* one atom per direction: out [B x M] = rotated atom cols[b] (the reference's rotate_atom /
 * interp_PGSE_from_multishell called with a single 1-D signal, mf_utils.py:1242-1243, 1764-1765) */
int mfx_rotate_cols(const mfx_plan* p, const double* dirs, const int32_t* cols, int64_t B, int normalise_dirs,
                    double* out);
int mfx_rotate_cols_dev(const mfx_plan* p, const double* d_dirs, const int32_t* d_cols, int64_t B,
                        int normalise_dirs, double* d_out, void* stream);

/* ---- the voxel loop: replaces MFModel.fit's loop over _fit_voxel
 * (mf.py:976-1032 calling mf.py:340-461).
 * Y       [V x M]           measured signals (ROI order)
 * K       [V]               number of fascicles per voxel (0..maxfasc), maxfasc <= 3 (MFModel.fit stops at 2, mf.py:467;
 *                           three fascicles - BASELINE config 5 - are opt-in here and run voxel by voxel through the
 *                           explicit-dictionary solver: reference solve_exhaustive_posweights_3 / _4up arithmetic)
 * csf,ear [V] or NULL       per-voxel compartment flags
 * peaks   [V x 3*maxfasc]   fascicle directions (unit norm +-1e-3, checked once per batch)
 * sig_csf [M] / sig_ear [M x E]  (NULL when csf_on / ear_on is 0)
 * params_out [V x num_params], num_params = 1 + 2*maxfasc + csf_on + 2*ear_on + 2,
 * layout of mf.py:375-450.                                                          */
int mfx_fit_batch(const mfx_plan* p, const double* Y, const int32_t* K, const uint8_t* csf, const uint8_t* ear,
                  const double* peaks, int maxfasc, int csf_on, int ear_on, const double* sig_csf,
                  const double* sig_ear, int E, int64_t V, double* params_out);
/* Same, with the reference's ROI gather `data[mask > 0]` (mf.py:644, 1020-1022) fused into the upload: voxel v's
 * signal is the M doubles at Y + rows[v] * M (rows == NULL: rows[v] = v).  Y is read chunk by chunk through pinned
 * staging buffers while the kernels of the previous chunks run (copy and compute streams).                    */
int mfx_fit_batch_rows(const mfx_plan* p, const double* Y, const int64_t* rows, const int32_t* K, const uint8_t* csf,
                       const uint8_t* ear, const double* peaks, int maxfasc, int csf_on, int ear_on,
                       const double* sig_csf, const double* sig_ear, int E, int64_t V, double* params_out);
/* Same, for a volume kept the way a NIfTI-1 file stores it (what nib.load(data) maps before get_fdata(), mf.py:623-626):
 * vol is [M][nvox] scalars of NIfTI data type vol_dtype (2 u8, 4 i16, 8 i32, 16 f32, 64 f64, 256 i8, 512 u16, 768 u32;
 * native byte order), one 3-D image per measurement; voxel v's signal is element vox[v] (0 <= vox[v] < nvox) of every
 * image.  The volume is uploaded as it is; the reference's get_fdata()[mask > 0] (mf.py:644: conversion to float64,
 * `x * scl_slope + scl_inter` when the header asks for it - scl_slope 0 means unscaled -, ROI gather) runs on the
 * device with the same two roundings.                                                                          */
int mfx_fit_batch_volume(const mfx_plan* p, const void* vol, int vol_dtype, double scl_slope, double scl_inter,
                         int64_t nvox, const int64_t* vox, const int32_t* K, const uint8_t* csf, const uint8_t* ear,
                         const double* peaks, int maxfasc, int csf_on, int ear_on, const double* sig_csf,
                         const double* sig_ear, int E, int64_t V, double* params_out);
/* The same conversion + gather for any other per-voxel quantity that comes as a file-order volume of ncomp components
 * (fascicle directions, tensors, colatitude / longitude: what MFModel.fit reads beside the data, mf.py:693-800, and
 * indexes with `[mask > 0]` like the data): rows_out [V x ncomp] float64 on the HOST.  Runs on `device`. */
int mfx_volume_rows(const void* vol, int vol_dtype, double scl_slope, double scl_inter, int64_t nvox, int ncomp,
                    const int64_t* vox, int64_t V, double* rows_out, int device);
/* mfx_fit_batch / mfx_fit_batch_rows / mfx_fit_batch_volume keep their pinned staging buffers and device buffers (signals, directions,
 * parameters) in the calling thread's state between calls and only ever grow them, so that a volume fitted slab by
 * slab (the reference's loop, mf.py:976-1032) pays no allocation per call; every entry point's scratch memory lives
 * in arenas the calling thread keeps per stream.  This returns all of it (after draining the thread's streams and
 * the devices its arenas are on); the next call allocates again.  No counterpart in the reference.            */
int mfx_thread_release(void);
/* Device-resident variant: all pointers are device pointers.  Only the homogeneous class
 * "every voxel has K == maxfasc, csf == csf_on, ear == ear_on" is accepted (that is what a
 * benchmark or a pre-binned caller has); mixed batches go through mfx_fit_batch.        */
int mfx_fit_batch_dev(const mfx_plan* p, const double* d_Y, const double* d_peaks, int maxfasc, int csf_on,
                      int ear_on, const double* d_sig_csf, const double* d_sig_ear, int E, int64_t V,
                      double* d_params_out, void* stream);

/* ---- explicit-dictionary solver: replaces solve_exhaustive_posweights(A, y, dicsizes)
 * (mf_utils.py:115-214 and the kernels it dispatches to, 225-657).
 * A [M x sum(dicsizes)] with leading dimension lda; outputs as the reference's 5-tuple. */
int mfx_solve_exhaustive(const double* A, int64_t lda, int M, const int64_t* dicsizes, int Kp, const double* y,
                         double* w, int64_t* sub, int64_t* tot, double* min_obj, double* y_rec);

/* ---- "next" row N3: the voxel loop of cleanup_2fascicles(frac1, frac2, peakmode, mu1, mu2, mask) (mf.py:36-335,
 * loop body mf.py:170-335).  Inputs are the ROI voxels (mask > 0) in np.where order with the orientation descriptors
 * already turned into direction vectors (the Python wrapper does that as the reference does: colat/longit trigonometry,
 * DT_vec_to_peaks for tensors): f1, f2 [n] weights, p1, p2 [n x 3] directions.  Per voxel, in the reference's order:
 * peaks closer than the merge angle (|clip(p1.p2)| > cos_min) are merged into population 0, a population `ratio` times
 * lighter than the other one and lighter than w_keep is dropped, populations lighter than w_small are dropped, the
 * survivors are ordered by descending weight.  peaks_out [n x 6], count_out [n] (0, 1 or 2 as doubles, like the
 * reference's num_fasc_out).  Results equal the reference's bit for bit.                                          */
int mfx_cleanup_2fascicles(const double* f1, const double* f2, const double* p1, const double* p2, int64_t n,
                           double cos_min, double ratio, double w_keep, double w_small, double* peaks_out,
                           double* count_out, int device);
int mfx_cleanup_2fascicles_dev(const double* d_f1, const double* d_f2, const double* d_p1, const double* d_p2, int64_t n,
                               double cos_min, double ratio, double w_keep, double w_small, double* d_peaks_out,
                               double* d_count_out, void* stream);

/* ---- Monte-Carlo signal synthesis (dictionary generation, upstream of fitting): replaces
 * monte_carlo_average(sim_phases, delta_mapping, gscaling, Dscaling, num_spins) (mf_utils.py:2758-2810),
 * the kernel under get_PGSE_from_phases (mf_utils.py:2813-3015).
 *   signal[i] = (1/num_spins) * sum_l cos(Dscaling * sum_n gscaling[i,n] * phases[delta_mapping[i]*num_spins + l, n])
 * sim_phases [n_entries x dim] row-major on the host (dim 1..3); delta_mapping int64[n_seq]; gscaling
 * [n_seq x dim] row-major; signal double[n_seq] (host).  MFX_ERR_ARG if a mapping points outside the
 * phase table (the reference would index out of bounds).                                            */
int mfx_monte_carlo_average(const double* sim_phases, int64_t n_entries, int dim, const int64_t* delta_mapping,
                            const double* gscaling, double Dscaling, int64_t num_spins, int64_t n_seq,
                            double* signal, int device);
/* Same with the phase table already on the current device, any layout: element (entry e, dimension d)
 * is d_phases[e*spin_stride + d*dim_stride] (row-major: dim,1; one plane per phase file: 1,n_entries).
 * delta_mapping, gscaling and signal stay host arrays (n_seq is small).                           */
int mfx_monte_carlo_average_dev(const double* d_phases, int64_t n_entries, int64_t spin_stride, int64_t dim_stride,
                                int dim, const int64_t* delta_mapping, const double* gscaling, double Dscaling,
                                int64_t num_spins, int64_t n_seq, double* signal, void* stream);

/* Timing hook for bench.py: average device time (ms) of the dominant kernel of the last
 * mfx_fit_batch*_ call on this thread, measured with hipEvents on the launch stream
 * (valid after the stream has been synchronised); < 0 if unavailable.              */
double mfx_last_kernel_ms(void);
void mfx_set_profiling(int enabled);
/* Diagnostic builds only (-DMFX_STAMPS): device buffer [grid x 16] of s_memtime stamps written by the
 * K=2 kernel at its phase boundaries; a no-op in the shipped build.                                */
void mfx_debug_set_stamps(void* dev_ptr);
/* Diagnostic: number of voxels of the last mfx_fit_batch* call on this thread that a fused kernel could not decide
 * from its short list and redid exactly: two-fascicle voxels the split-FP16 screening kernel handed back to the FP64
 * kernel (ring overflow or screening-error guard), and two-fascicle + CSF/EAR voxels that took the exhaustive exact
 * pass; 0 in the normal case.  The counters are summed on the device and copied behind the kernels: this call waits
 * for that copy (the fit calls themselves never do).  ..._guard_count: how many of them the screening-error guard sent. */
int mfx_debug_last_fallback_count(void);
int mfx_debug_last_guard_count(void);
/* Diagnostic: raw counter `which` of the last call: 0 hand-backs / exhaustive passes, 1 guard hand-backs, and for the
 * two-fascicle + CSF/EAR kernel 2 short-listed pairs, 3 family items (one-atom / no-atom supports and ambiguous slots
 * evaluated exactly), 4 voxels of the [N, N, 1] screening pipeline handed to the FP64 kernel of the class (ring overflow,
 * an atom inside the span of the CSF column, list overflow, bound check), 5 of them by the bound check (a listed pair
 * whose exact score exceeds its screening bound by more than the margin), summed over the batch; and the screening
 * kernels' population audit (every two-fascicle voxel compares the split-FP16 cross product of ONE pseudo-random atom pair
 * with its FP64 value, listed or not): 8 audited pairs whose error exceeds a quarter of the screening margin, 9 the largest
 * error in units of 1e-11 (cosine units), 10 audited pairs; which = 0..11. */
int mfx_debug_last_counter(int which);
/* Diagnostic: candidate-list entries per voxel of the batched three-fascicle path (fit_k3.hip; 0: the built-in 4 M).  A voxel
 * whose list overflows is redone by the voxel-by-voxel path, gated on the device; tests lower the cap to force that. */
void mfx_debug_set_k3_cap(int cap);
/* Diagnostic: 1 sends every voxel class (0, 1 or 2 fascicles, with or without CSF / EAR) of mfx_fit_batch* voxel by voxel
 * through the explicit-dictionary solver, the path shapes beyond the fused kernels' limits take by themselves (dictionaries
 * or protocols too large for the LDS, more than 16 CSF+EAR columns); tests use it to check that path on small shapes. */
void mfx_debug_set_force_generic(int enabled);
/* Diagnostic: 0 sends problems with three sub-dictionaries (mfx_fit_batch* with three fascicles, mfx_solve_exhaustive) through
 * the plain scan - one thread per index triple, every triple scored, no batched path and no relaxed-bound screen - the
 * unscreened referee of the full-size three-fascicle test (about 40 ms per voxel at 1 500 atoms). */
void mfx_debug_set_k3_screen(int enabled);
/* Diagnostic: 0 routes two-fascicle voxels to the FP64 kernel only (same as MFX_K2_SCREEN=0 in the environment). */
void mfx_debug_set_k2_screen(int enabled);
/* Diagnostic: 1 routes two-fascicle voxels of protocols with 129..256 measurements to the wide (one wave per SIMD)
 * screening kernel instead of the two-waves-per-SIMD one, 0 never uses the wide kernel (protocols of more than 256
 * measurements then run on the FP64 kernel), any other value: automatic (same as MFX_K2_WIDE in the environment). */
void mfx_debug_set_k2_wide(int mode);
/* 0 keeps the two-fascicle + CSF class ([N, N, 1]) on the FP64 kernel instead of the screening pipeline (split-FP16
 * screening kernel -> short lists -> exact stage); same as MFX_K2X_SCREEN=0 in the environment.  Calling thread only. */
void mfx_debug_set_k2x_screen(int on);
/* Diagnostic: short-list size of the FP64 two-fascicle kernel beyond which its exhaustive exact pass runs
 * (default and maximum 256; 0 forces that pass for every voxel -- used by the tests to cover it). */
void mfx_debug_set_k2_maxc(int maxc);
/* Diagnostic: the same for the two-fascicle + CSF/EAR kernel (default and maximum 256; 0 forces its exhaustive pass). */
void mfx_debug_set_k2x_maxc(int maxc);
/* Diagnostic: number of short-list ring entries the screening kernel uses (rounded down to a power of two;
 * default and maximum 2048, <= 0 restores the default) -- the tests lower it to force hand-backs to the FP64 kernel. */
void mfx_debug_set_k2s_cap(int cap);
/* Diagnostic: 2 forces the screening kernel's two-image schedule (a workgroup barrier per half-step) that otherwise
 * serves only the shapes whose LDS footprint leaves no room for a third chunk image; any other value: automatic. */
void mfx_debug_set_k2s_images(int nb);

#ifdef __cplusplus
}
#endif
#endif
