// C ABI of libgama_vtm.so (see include/gama_vtm.h).  Host side only: validation, table
// design/upload, launches.  No exception crosses the extern "C" frame (the reference's
// plugin convention: construct returns nullptr, VocalTractModelPlugin.cpp:87-90).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/gama_vtm.h"
#include "vtm_design.hpp"
#include "vtm_kernels.hpp"
#include "vtm_tracks.hpp"
#include "vtm_math.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int status, const std::string& msg)
{
	g_last_error = msg;
	return status;
}

int fail_hip(hipError_t e, const char* what)
{
	return fail(GVTM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

struct DeviceBuffer {
	void* ptr = nullptr;
	size_t bytes = 0;
	hipError_t ensure(size_t need)
	{
		if (need <= bytes) return hipSuccess;
		if (ptr) (void) hipFree(ptr);
		ptr = nullptr;
		bytes = 0;
		hipError_t e = hipMalloc(&ptr, need);
		if (e == hipSuccess) bytes = need;
		return e;
	}
	void release()
	{
		if (ptr) (void) hipFree(ptr);
		ptr = nullptr;
		bytes = 0;
	}
};

struct EventPair {
	hipEvent_t start = nullptr, stop = nullptr;
};

// Makes the plan's device current for the duration of an entry point and gives the caller's device back afterwards.
class DeviceScope {
public:
	explicit DeviceScope(int device)
	{
		if (hipGetDevice(&previous_) != hipSuccess) previous_ = -1;
		status_ = (previous_ == device) ? hipSuccess : hipSetDevice(device);
		changed_ = status_ == hipSuccess && previous_ != device;
	}
	~DeviceScope()
	{
		if (changed_ && previous_ >= 0) (void) hipSetDevice(previous_);
	}
	DeviceScope(const DeviceScope&) = delete;
	DeviceScope& operator=(const DeviceScope&) = delete;
	hipError_t status() const { return status_; }
private:
	int previous_ = -1;
	bool changed_ = false;
	hipError_t status_ = hipSuccess;
};

} // namespace

#ifndef GVTM_NOISE_TABLE
#define GVTM_NOISE_TABLE 1
#endif

struct gvtm_plan {
	gvtm::Design design;
	int device = 0;
	int precision = GVTM_PRECISION_F64;
	int rows = 0;       // utterances per workgroup; 0 = by batch size (a diagnostics build can force it)
	// design tables on the device: double, or float (as designed) for GVTM_PRECISION_F32
	void* d_wavetable = nullptr;
	void* d_fir = nullptr;
	void* d_src_h = nullptr;
	void* d_src_dh = nullptr;
	// the noise source's samples by internal step (the same for every utterance), grown on demand; superseded buffers stay
	// allocated until the plan goes (a launch in flight on the caller's stream may still read them)
	void* d_noise = nullptr;
	size_t noise_len = 0;
	std::vector<void*> noise_retired;
	gvtm::DeviceConstants* d_consts = nullptr;
	gvtm::Model5Constants* d_consts5 = nullptr; // model 5 plans only
	// staging for the host-buffer entry point
	DeviceBuffer s_params, s_frames, s_audio, s_counts, s_maxabs, s_pcm, s_scales;
	int compute_units = 0; // of the plan's device (the host entries cut big batches into slices that fill them once)
	// kernel timing (HIP events on the launch stream)
	bool timing = false;
	double* debug_taps = nullptr; // device pointer (diagnostics builds: gvtm_debug_set_taps)
	unsigned long long* phase_cycles = nullptr; // device pointer (diagnostics builds: gvtm_debug_set_phase_cycles)
	std::vector<EventPair> pending;
	std::vector<EventPair> pool;
	// host-buffer entries: frames in on one stream, kernels on a second, samples out on a third (created on first use)
	hipStream_t h2d_stream = nullptr, compute_stream = nullptr, copy_stream = nullptr;
	std::vector<hipEvent_t> slice_done; // two per slice: frames arrived, samples ready
	// model 5: utterances per workgroup.  One, whatever the batch: the two-utterance shape (two tube wavefronts, chunk of 24
	// steps -- what LDS holds of the 62-entry tube records) measured SLOWER at every batch size (batch 512 x 250 frames:
	// 17.3 ms against 2 x 6.97 ms; profiles/r03_role_cycles_m5.txt): the passes are latency-bound, so a chunk of 24 steps
	// costs what one of 60 does, and the tube wavefronts slow down from 268 to 430 cycles per step next to five busy
	// helpers.  A diagnostics build can still force it (tests hold it to the one-utterance shape's samples bit for bit).
	int rows5_for(size_t) const { return rows == 2 ? 2 : 1; }
};

namespace {

template <typename P, typename T>
hipError_t upload(P** dst, const std::vector<T>& src)
{
	hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), sizeof(T) * src.size());
	if (e != hipSuccess) return e;
	return hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice);
}

void free_plan(gvtm_plan* p)
{
	if (!p) return;
	if (p->device == GVTM_DEVICE_NONE) {
		delete p;
		return;
	}
	DeviceScope scope(p->device);
	if (p->d_wavetable) (void) hipFree(p->d_wavetable);
	if (p->d_fir) (void) hipFree(p->d_fir);
	if (p->d_src_h) (void) hipFree(p->d_src_h);
	if (p->d_src_dh) (void) hipFree(p->d_src_dh);
	if (p->d_noise) (void) hipFree(p->d_noise);
	for (void* q : p->noise_retired) (void) hipFree(q);
	if (p->d_consts) (void) hipFree(p->d_consts);
	if (p->d_consts5) (void) hipFree(p->d_consts5);
	p->s_params.release();
	p->s_frames.release();
	p->s_audio.release();
	p->s_counts.release();
	p->s_maxabs.release();
	p->s_pcm.release();
	p->s_scales.release();
	if (p->h2d_stream) (void) hipStreamDestroy(p->h2d_stream);
	if (p->compute_stream) (void) hipStreamDestroy(p->compute_stream);
	if (p->copy_stream) (void) hipStreamDestroy(p->copy_stream);
	for (hipEvent_t ev : p->slice_done) (void) hipEventDestroy(ev);
	for (auto& ev : p->pending) { (void) hipEventDestroy(ev.start); (void) hipEventDestroy(ev.stop); }
	for (auto& ev : p->pool) { (void) hipEventDestroy(ev.start); (void) hipEventDestroy(ev.stop); }
	delete p;
}

} // namespace

extern "C" {

const char* gvtm_status_string(int status)
{
	switch (status) {
	case GVTM_OK: return "ok";
	case GVTM_ERR_INVALID_ARGUMENT: return "invalid argument";
	case GVTM_ERR_NO_DEVICE: return "no HIP device";
	case GVTM_ERR_HIP: return "HIP runtime error";
	case GVTM_ERR_UNSUPPORTED: return "unsupported";
	case GVTM_ERR_OUT_OF_MEMORY: return "out of memory";
	default: return "unknown status";
	}
}

const char* gvtm_last_error(void)
{
	return g_last_error.c_str();
}

int gvtm_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int gvtm_plan_create(const gvtm_config* config, double control_rate, int device, gvtm_plan** plan_out)
{
	if (!config || !plan_out) return fail(GVTM_ERR_INVALID_ARGUMENT, "null config or plan_out");
	*plan_out = nullptr;
	try {
		std::unique_ptr<gvtm_plan, void (*)(gvtm_plan*)> plan(new gvtm_plan, free_plan);
		const std::string why = gvtm::design_plan(*config, control_rate, plan->design);
		if (!why.empty()) return fail(GVTM_ERR_INVALID_ARGUMENT, why);
		plan->precision = config->precision;

		if (device == GVTM_DEVICE_NONE) {
			// design-only plan: info, tables and output counts work, synthesis reports NO_DEVICE
			plan->device = GVTM_DEVICE_NONE;
			*plan_out = plan.release();
			return GVTM_OK;
		}
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) {
			return fail(GVTM_ERR_NO_DEVICE, "no HIP device available (libgama_vtm has no CPU path)");
		}
		if (device < 0 || device >= n) return fail(GVTM_ERR_NO_DEVICE, "device index out of range");
		plan->device = device;
		DeviceScope scope(device);
		if ((e = scope.status()) != hipSuccess) return fail_hip(e, "hipSetDevice");
		hipDeviceProp_t prop;
		if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail_hip(e, "hipGetDeviceProperties");
		if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
			return fail(GVTM_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
		}
		plan->compute_units = prop.multiProcessorCount;
		const gvtm::Design& dg = plan->design;
		if (dg.f32) {
			if ((e = upload(&plan->d_wavetable, dg.wavetable_f)) != hipSuccess) return fail_hip(e, "upload wavetable");
			if ((e = upload(&plan->d_fir, dg.fir_f)) != hipSuccess) return fail_hip(e, "upload fir");
			if ((e = upload(&plan->d_src_h, dg.src_h_f)) != hipSuccess) return fail_hip(e, "upload src_h");
			if ((e = upload(&plan->d_src_dh, dg.src_dh_f)) != hipSuccess) return fail_hip(e, "upload src_dh");
		} else {
			if ((e = upload(&plan->d_wavetable, dg.wavetable)) != hipSuccess) return fail_hip(e, "upload wavetable");
			if ((e = upload(&plan->d_fir, dg.fir)) != hipSuccess) return fail_hip(e, "upload fir");
			if ((e = upload(&plan->d_src_h, dg.src_h)) != hipSuccess) return fail_hip(e, "upload src_h");
			if ((e = upload(&plan->d_src_dh, dg.src_dh)) != hipSuccess) return fail_hip(e, "upload src_dh");
		}
		if ((e = upload(&plan->d_consts, std::vector<gvtm::DeviceConstants>(1, plan->design.k))) != hipSuccess) return fail_hip(e, "upload constants");
		*plan_out = plan.release();
		return GVTM_OK;
	} catch (const std::bad_alloc&) {
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	} catch (const std::exception& ex) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, ex.what());
	}
}

int gvtm_plan_create_model5(const gvtm5_config* config, double control_rate, int device, gvtm_plan** plan_out)
{
	if (!config || !plan_out) return fail(GVTM_ERR_INVALID_ARGUMENT, "null config or plan_out");
	*plan_out = nullptr;
	try {
		std::unique_ptr<gvtm_plan, void (*)(gvtm_plan*)> plan(new gvtm_plan, free_plan);
		const std::string why = gvtm::design_plan5(*config, control_rate, plan->design);
		if (!why.empty()) return fail(GVTM_ERR_INVALID_ARGUMENT, why);
		plan->precision = GVTM_PRECISION_F64;
		if (device == GVTM_DEVICE_NONE) {
			plan->device = GVTM_DEVICE_NONE;
			*plan_out = plan.release();
			return GVTM_OK;
		}
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) return fail(GVTM_ERR_NO_DEVICE, "no HIP device available (libgama_vtm has no CPU path)");
		if (device < 0 || device >= n) return fail(GVTM_ERR_NO_DEVICE, "device index out of range");
		plan->device = device;
		DeviceScope scope(device);
		if ((e = scope.status()) != hipSuccess) return fail_hip(e, "hipSetDevice");
		hipDeviceProp_t prop;
		if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail_hip(e, "hipGetDeviceProperties");
		if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
			return fail(GVTM_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
		}
		plan->compute_units = prop.multiProcessorCount;
		const gvtm::Design& dg = plan->design;
		if ((e = upload(&plan->d_src_h, dg.src_h)) != hipSuccess) return fail_hip(e, "upload src_h");
		if ((e = upload(&plan->d_src_dh, dg.src_dh)) != hipSuccess) return fail_hip(e, "upload src_dh");
		if ((e = upload(&plan->d_consts, std::vector<gvtm::DeviceConstants>(1, dg.k))) != hipSuccess) return fail_hip(e, "upload constants");
		if ((e = upload(&plan->d_consts5, std::vector<gvtm::Model5Constants>(1, dg.k5))) != hipSuccess) return fail_hip(e, "upload model 5 constants");
		*plan_out = plan.release();
		return GVTM_OK;
	} catch (const std::bad_alloc&) {
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	} catch (const std::exception& ex) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, ex.what());
	}
}

void gvtm_plan_destroy(gvtm_plan* plan)
{
	free_plan(plan);
}

int gvtm_plan_info(const gvtm_plan* plan, gvtm_info* info)
{
	if (!plan || !info) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan or info");
	const gvtm::DeviceConstants& k = plan->design.k;
	info->internal_sample_rate = k.sample_rate;
	info->control_steps = k.control_steps;
	info->output_rate = plan->design.config.output_rate;
	info->control_rate = plan->design.control_rate;
	info->fir_taps = k.fir_taps;
	info->time_register_increment = k.time_inc;
	info->phase_increment = k.phase_inc;
	info->pad_size = k.pad;
	info->upsampling = k.upsampling;
	info->device = plan->device;
	info->precision = plan->precision;
	info->section_delay = k.section_delay;
	info->model5 = plan->design.model5 ? 1 : 0;
	info->reserved_ = 0;
	info->internal_rate_hz = plan->design.model5 ? plan->design.k5.sample_rate : static_cast<double>(k.sample_rate);
	return GVTM_OK;
}

int gvtm_plan_table(const gvtm_plan* plan, int which, double* out, size_t capacity)
{
	if (!plan || !out) return -fail(GVTM_ERR_INVALID_ARGUMENT, "null plan or out");
	const std::vector<double>* src = nullptr;
	switch (which) {
	case GVTM_TABLE_FIR: src = &plan->design.fir; break;
	case GVTM_TABLE_SRC_H: src = &plan->design.src_h; break;
	case GVTM_TABLE_SRC_DH: src = &plan->design.src_dh; break;
	case GVTM_TABLE_WAVETABLE: src = &plan->design.wavetable; break;
	default: return -fail(GVTM_ERR_INVALID_ARGUMENT, "unknown table");
	}
	if (src->empty()) return -fail(GVTM_ERR_INVALID_ARGUMENT, "this model has no such table");
	if (capacity < src->size()) return -fail(GVTM_ERR_INVALID_ARGUMENT, "table buffer too small");
	std::memcpy(out, src->data(), sizeof(double) * src->size());
	return static_cast<int>(src->size());
}

size_t gvtm_output_count(const gvtm_plan* plan, size_t n_frames)
{
	if (!plan) return static_cast<size_t>(-1);
	const gvtm::DeviceConstants& k = plan->design.k;
	return static_cast<size_t>(gvtm::src_output_count(k.time_inc, k.pad, k.upsampling, static_cast<uint64_t>(n_frames) * k.control_steps));
}

size_t gvtm_output_capacity(const gvtm_plan* plan, size_t max_frames)
{
	if (!plan) return static_cast<size_t>(-1);
	const gvtm::DeviceConstants& k = plan->design.k;
	return static_cast<size_t>(gvtm::src_output_capacity(k.time_inc, k.pad, k.upsampling, static_cast<uint64_t>(max_frames) * k.control_steps));
}

#ifdef GVTM_DIAGNOSTICS
/* ---- hooks of the diagnostics build (libgama_vtm_diag.so; tests and tools only, not in the public header) ---- */
#pragma GCC visibility push(default)

/* Forces the utterances per workgroup (1, 2, 4, 8; 0 = by batch size), i.e. the kernel shape a big batch would get. */
int gvtm_debug_set_rows(gvtm_plan* plan, int rows)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	plan->rows = rows;
	return GVTM_OK;
}

/* LDS bytes of a workgroup of `rows` utterances of this plan's one-shot kernel (0 = model 5's). */
size_t gvtm_debug_lds_bytes(const gvtm_plan* plan, int rows)
{
	if (!plan) return 0;
	if (plan->design.model5) return gvtm::synth5_lds_bytes(rows == 2 ? 2 : 1);
	return gvtm::synth_lds_bytes(plan->design.k, plan->precision, rows, 0);
}

/* The plan-level noise-sample table as the host builds it (n floats or doubles); needs no device. */
int gvtm_debug_noise_table(size_t n, int as_float, void* out)
{
	if (!out) return fail(GVTM_ERR_INVALID_ARGUMENT, "null buffer");
	gvtm::design_noise_table(n, as_float != 0, out);
	return GVTM_OK;
}

/* Test hook (not in the public header): device buffer [batch][max_frames*control_steps][8] of
 * doubles that receives per-step intermediate values of the next synthesis calls; null disables. */
int gvtm_debug_set_taps(gvtm_plan* plan, double* d_taps)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	plan->debug_taps = d_taps;
	return GVTM_OK;
}

/* Diagnostic hook (not in the public header): device buffer [batch][16] of uint64 receiving the
 * shader cycles workgroup `b` spent per phase (generation 1) or per role wavefront [0..7] and per
 * helper stage [8..13] (generation 2). */
int gvtm_debug_set_phase_cycles(gvtm_plan* plan, unsigned long long* d_cycles)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	plan->phase_cycles = d_cycles;
	return GVTM_OK;
}

/* Diagnostic hook: what each lane receives through the cross-lane primitives of the tube
 * wavefront (row_shr:1, row_shl:1, row_ror:6, row_ror:10, wave_shr:1, wave_shl:1); out[6][64] host ints. */
int gvtm_debug_dpp_selftest(gvtm_plan* plan, int* out)
{
	if (!plan || !out || plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_INVALID_ARGUMENT, "needs a device plan");
	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	int* d = nullptr;
	if ((e = hipMalloc(reinterpret_cast<void**>(&d), gvtm::kDppSelftestInts * sizeof(int))) != hipSuccess) return fail_hip(e, "hipMalloc");
	e = gvtm::launch_dpp_selftest(d, nullptr);
	if (e == hipSuccess) e = hipMemcpy(out, d, gvtm::kDppSelftestInts * sizeof(int), hipMemcpyDeviceToHost);
	(void) hipFree(d);
	if (e != hipSuccess) return fail_hip(e, "dpp selftest");
	return GVTM_OK;
}

/* Test hook: the short elementary functions of csrc/vtm_math.hpp evaluated on the host
 * (kind 0 = 2^x, 1 = 10^x, 2 = cos, 3 = tan; 4 = powf(2, x), 5 = powf(10, x), 6 = cosf, 7 = tanf of the
 * all-float path, arguments and results carried as doubles). */
int gvtm_debug_short_math(int kind, const double* x, size_t n, double* out)
{
	if (!x || !out) return GVTM_ERR_INVALID_ARGUMENT;
	for (size_t i = 0; i < n; ++i) {
		switch (kind) {
		case 0: out[i] = gvtm::vmath::exp2_short(x[i]); break;
		case 1: out[i] = gvtm::vmath::exp10_short(x[i]); break;
		case 2: out[i] = gvtm::vmath::cos_short(x[i]); break;
		case 3: out[i] = gvtm::vmath::tan_short(x[i]); break;
		case 4: out[i] = gvtm::vmath::powf_base2(static_cast<float>(x[i])); break;
		case 5: out[i] = gvtm::vmath::powf_base10(static_cast<float>(x[i])); break;
		case 6: out[i] = gvtm::vmath::cosf_glibc(static_cast<float>(x[i])); break;
		case 7: out[i] = gvtm::vmath::tanf_glibc(static_cast<float>(x[i])); break;
		default: return GVTM_ERR_INVALID_ARGUMENT;
		}
	}
	return GVTM_OK;
}

/* Test hook: Util::frequency / Util::amplitude60dB / tan / cos of the all-float path evaluated by a
 * kernel on the plan's device (kind 0..3), host arrays in and out. */
int gvtm_debug_device_float_math(gvtm_plan* plan, int kind, const float* x, size_t n, float* out)
{
	if (!plan || !x || !out || kind < 0 || kind > 4) return fail(GVTM_ERR_INVALID_ARGUMENT, "bad argument");
	if (plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_NO_DEVICE, "design-only plan");
	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	float *dx = nullptr, *dout = nullptr;
	if ((e = hipMalloc(reinterpret_cast<void**>(&dx), n * sizeof(float))) != hipSuccess) return fail_hip(e, "hipMalloc");
	if ((e = hipMalloc(reinterpret_cast<void**>(&dout), n * sizeof(float))) != hipSuccess) { (void) hipFree(dx); return fail_hip(e, "hipMalloc"); }
	e = hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = gvtm::launch_float_math_probe(kind, dx, n, dout, nullptr);
	if (e == hipSuccess) e = hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost);
	(void) hipFree(dx);
	(void) hipFree(dout);
	if (e != hipSuccess) return fail_hip(e, "float math probe");
	return GVTM_OK;
}

#pragma GCC visibility pop
#endif /* GVTM_DIAGNOSTICS */

size_t gvtm_tracks_frame_count(const gvtm_track_config* config, const gvtm_event* events, size_t n_events)
{
	if (!config || (!events && n_events > 0)) { fail(GVTM_ERR_INVALID_ARGUMENT, "null config or events"); return static_cast<size_t>(-1); }
	gvtm::TrackConstants k;
	const char* why = gvtm::design_tracks(*config, k);
	if (why[0]) { fail(GVTM_ERR_INVALID_ARGUMENT, why); return static_cast<size_t>(-1); }
	return gvtm::tracks_frame_count(k.control_period, events, n_events);
}

int gvtm_generate_tracks_device(int device, const gvtm_track_config* config, const gvtm_event* d_events,
		const int64_t* d_event_offsets, size_t batch, size_t max_frames, float* d_params, int32_t* d_frame_counts,
		gvtm_drift_state* d_drift, void* hip_stream)
{
	if (!config || !d_event_offsets || (!d_params && max_frames > 0)) return fail(GVTM_ERR_INVALID_ARGUMENT, "null argument");
	gvtm::TrackArgs args{};
	const char* why = gvtm::design_tracks(*config, args.k);
	if (why[0]) return fail(GVTM_ERR_INVALID_ARGUMENT, why);
	if (batch == 0) return GVTM_OK;
	if (!d_events) return fail(GVTM_ERR_INVALID_ARGUMENT, "null events");
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(GVTM_ERR_NO_DEVICE, "no HIP device available (track generation has no CPU path)");
	if (device < 0 || device >= n) return fail(GVTM_ERR_NO_DEVICE, "device index out of range");
	DeviceScope scope(device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	args.events = d_events;
	args.event_offsets = d_event_offsets;
	args.batch = batch;
	args.max_frames = max_frames;
	args.params = d_params;
	args.frame_counts = d_frame_counts;
	args.drift = d_drift;
	e = gvtm::launch_tracks(args, static_cast<hipStream_t>(hip_stream));
	if (e != hipSuccess) return fail_hip(e, "track generation launch");
	return GVTM_OK;
}

int gvtm_generate_tracks_host(int device, const gvtm_track_config* config, const gvtm_event* events,
		const int64_t* event_offsets, size_t batch, size_t max_frames, float* params, int32_t* frame_counts,
		gvtm_drift_state* drift)
{
	if (!config || !event_offsets || (!params && max_frames > 0)) return fail(GVTM_ERR_INVALID_ARGUMENT, "null argument");
	if (batch == 0) return GVTM_OK;
	if (!events) return fail(GVTM_ERR_INVALID_ARGUMENT, "null events");
	for (size_t b = 0; b < batch; ++b) {
		if (event_offsets[b + 1] < event_offsets[b] || event_offsets[0] != 0) return fail(GVTM_ERR_INVALID_ARGUMENT, "event_offsets must start at 0 and not decrease");
	}
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(GVTM_ERR_NO_DEVICE, "no HIP device available (track generation has no CPU path)");
	if (device < 0 || device >= n) return fail(GVTM_ERR_NO_DEVICE, "device index out of range");
	DeviceScope scope(device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	const size_t n_events = static_cast<size_t>(event_offsets[batch]);
	DeviceBuffer d_ev, d_off, d_par, d_cnt, d_dr;
	int rc = GVTM_OK;
	auto done = [&](int code) { d_ev.release(); d_off.release(); d_par.release(); d_cnt.release(); d_dr.release(); return code; };
	if ((e = d_ev.ensure(sizeof(gvtm_event) * (n_events ? n_events : 1))) != hipSuccess) return done(fail_hip(e, "hipMalloc"));
	if ((e = d_off.ensure(sizeof(int64_t) * (batch + 1))) != hipSuccess) return done(fail_hip(e, "hipMalloc"));
	if ((e = d_par.ensure(sizeof(float) * 16 * (batch * max_frames ? batch * max_frames : 1))) != hipSuccess) return done(fail_hip(e, "hipMalloc"));
	if ((e = d_cnt.ensure(sizeof(int32_t) * batch)) != hipSuccess) return done(fail_hip(e, "hipMalloc"));
	if (drift && (e = d_dr.ensure(sizeof(gvtm_drift_state) * batch)) != hipSuccess) return done(fail_hip(e, "hipMalloc"));
	if (n_events && (e = hipMemcpy(d_ev.ptr, events, sizeof(gvtm_event) * n_events, hipMemcpyHostToDevice)) != hipSuccess) return done(fail_hip(e, "H2D events"));
	if ((e = hipMemcpy(d_off.ptr, event_offsets, sizeof(int64_t) * (batch + 1), hipMemcpyHostToDevice)) != hipSuccess) return done(fail_hip(e, "H2D offsets"));
	if (drift && (e = hipMemcpy(d_dr.ptr, drift, sizeof(gvtm_drift_state) * batch, hipMemcpyHostToDevice)) != hipSuccess) return done(fail_hip(e, "H2D drift"));
	if ((e = hipMemset(d_par.ptr, 0, sizeof(float) * 16 * batch * max_frames)) != hipSuccess) return done(fail_hip(e, "hipMemset"));
	rc = gvtm_generate_tracks_device(device, config, static_cast<const gvtm_event*>(d_ev.ptr), static_cast<const int64_t*>(d_off.ptr), batch,
			max_frames, static_cast<float*>(d_par.ptr), static_cast<int32_t*>(d_cnt.ptr), drift ? static_cast<gvtm_drift_state*>(d_dr.ptr) : nullptr, nullptr);
	if (rc != GVTM_OK) return done(rc);
	if ((e = hipDeviceSynchronize()) != hipSuccess) return done(fail_hip(e, "track generation kernel"));
	if (params && (e = hipMemcpy(params, d_par.ptr, sizeof(float) * 16 * batch * max_frames, hipMemcpyDeviceToHost)) != hipSuccess) return done(fail_hip(e, "D2H frames"));
	if (frame_counts && (e = hipMemcpy(frame_counts, d_cnt.ptr, sizeof(int32_t) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return done(fail_hip(e, "D2H counts"));
	if (drift && (e = hipMemcpy(drift, d_dr.ptr, sizeof(gvtm_drift_state) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return done(fail_hip(e, "D2H drift"));
	return done(GVTM_OK);
}

int gvtm_plan_set_timing(gvtm_plan* plan, int enabled)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	plan->timing = enabled != 0;
	return GVTM_OK;
}

double gvtm_plan_take_kernel_ms(gvtm_plan* plan, int* launches_out)
{
	if (launches_out) *launches_out = 0;
	if (!plan || plan->device == GVTM_DEVICE_NONE || plan->pending.empty()) return -1.0;
	DeviceScope scope(plan->device);
	double total = 0.0;
	int n = 0;
	for (auto& ev : plan->pending) {
		float ms = 0.f;
		if (hipEventSynchronize(ev.stop) == hipSuccess && hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess) {
			total += ms;
			++n;
		}
		plan->pool.push_back(ev);
	}
	plan->pending.clear();
	if (launches_out) *launches_out = n;
	return n ? total / n : -1.0;
}

} // extern "C"

namespace {

// what a launch on behalf of a stream adds to the one-shot launch
struct StreamLaunch {
	unsigned char* d_state;
	size_t stride;
	int mode;     // gvtm::StreamMode
	int xr;       // the stream's ring length (one for all shapes)
	int rows;     // 1 unless the utterances are in lockstep
};

int launch_batch(gvtm_plan* plan, const float* d_params, const int32_t* d_frame_counts,
		size_t batch, size_t max_frames, float* d_audio, size_t audio_stride,
		int64_t* d_out_counts, float* d_maxabs, void* hip_stream, const StreamLaunch* sl)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	if (plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_NO_DEVICE, "design-only plan (GVTM_DEVICE_NONE): there is no CPU synthesis path");
	if (batch == 0) return GVTM_OK;
	if (!d_audio) return fail(GVTM_ERR_INVALID_ARGUMENT, "null audio buffer");
	if (max_frames > 0 && !d_params) return fail(GVTM_ERR_INVALID_ARGUMENT, "null params with max_frames > 0");
	if (batch > 0x7fffffffu) return fail(GVTM_ERR_INVALID_ARGUMENT, "batch too large for one launch");
	if (static_cast<unsigned long long>(max_frames) * plan->design.k.control_steps + 4096ull >= (1ull << 31)) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, "max_frames * control_steps does not fit the 31-bit step counter");
	}
	if (!sl && audio_stride < gvtm_output_count(plan, max_frames)) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, "audio_stride smaller than gvtm_output_count(plan, max_frames)");
	}
	const bool model5 = plan->design.model5;
	const gvtm::DeviceConstants& k = plan->design.k;
	constexpr size_t kLdsPerWorkgroup = 160 * 1024;
	int rows = model5 ? plan->rows5_for(batch) : gvtm::synth_rows(plan->precision, batch, plan->rows, k.section_delay);
	if (sl && sl->rows == 1) rows = 1;
	const int xr_fixed = sl ? sl->xr : 0;
	// a shape whose rings do not fit (down-sampling plans carry the reference's 1024-sample ring per row) gives way to
	// the next smaller one
	while (!model5 && rows > 1 && gvtm::synth_lds_bytes(k, plan->precision, rows, xr_fixed) > kLdsPerWorkgroup) rows /= 2;
	if ((model5 ? gvtm::synth5_lds_bytes(rows) : gvtm::synth_lds_bytes(k, plan->precision, rows, xr_fixed)) > kLdsPerWorkgroup) {
		return fail(GVTM_ERR_UNSUPPORTED, "LDS budget exceeded");
	}

	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);

	gvtm::SynthArgs args;
	args.k = k;
	args.kconst = plan->d_consts;
	args.params = d_params;
	args.frame_counts = d_frame_counts;
	args.audio = d_audio;
	args.out_counts = d_out_counts;
	args.maxabs = d_maxabs;
	args.wavetable = plan->d_wavetable;
	args.fir = plan->d_fir;
	std::memset(&args.fir_k, 0, sizeof(args.fir_k));
	if (!model5) {
		const gvtm::Design& dg = plan->design;
		if (dg.f32) {
			for (size_t i = 0; i < dg.fir_f.size() && i < 64; ++i) args.fir_k.f[i] = dg.fir_f[i];
		} else {
			for (size_t i = 0; i < dg.fir.size() && i < 49; ++i) args.fir_k.d[i] = dg.fir[i];
		}
	}
	args.src_h = plan->d_src_h;
	args.src_dh = plan->d_src_dh;
	if (!model5 && !sl && GVTM_NOISE_TABLE && plan->precision == GVTM_PRECISION_F32) {
		// one-shot launches read the noise samples from the plan's table (streams generate them: their length has no bound)
		const size_t steps = max_frames * static_cast<size_t>(k.control_steps);
		constexpr size_t kNoiseTableMaxSteps = size_t(1) << 26; // 256 MB of table (~55 min of audio per utterance): beyond it the kernel generates the samples
		if (steps > kNoiseTableMaxSteps) {
			// (args.noise_lp stays null)
		} else if (steps > plan->noise_len) {
			// geometric growth (at least twice the old table, in units of 2^18 steps): a caller whose lengths creep up
			// retires at most eight tables on the way to the cap
			size_t want = ((steps + (size_t(1) << 18) - 1) >> 18) << 18;
			want = std::min(std::max(want, 2 * plan->noise_len), kNoiseTableMaxSteps);
			const bool f32 = true; // (the double paths generate the samples in the kernel: measured faster there)
			std::vector<unsigned char> host(want * (f32 ? sizeof(float) : sizeof(double)));
			gvtm::design_noise_table(want, f32, host.data());
			void* fresh = nullptr;
			if ((e = hipMalloc(&fresh, host.size())) != hipSuccess) return fail_hip(e, "hipMalloc (noise table)");
			if ((e = hipMemcpy(fresh, host.data(), host.size(), hipMemcpyHostToDevice)) != hipSuccess) {
				(void) hipFree(fresh);
				return fail_hip(e, "hipMemcpy (noise table)");
			}
			if (plan->d_noise) plan->noise_retired.push_back(plan->d_noise);
			plan->d_noise = fresh;
			plan->noise_len = want;
		}
		if (steps <= kNoiseTableMaxSteps) {
			args.noise_lp = plan->d_noise;
			args.noise_len = plan->noise_len;
		}
	}
	args.max_frames = max_frames;
	args.audio_stride = audio_stride;
	args.batch = batch;
	args.xr = model5 ? 0 : (sl ? sl->xr : gvtm::synth_ring_length(k, plan->precision, rows));
	if (sl) {
		args.stream = sl->d_state;
		args.stream_stride = sl->stride;
		args.stream_mode = sl->mode;
	}
	args.debug_taps = plan->debug_taps;
	args.phase_cycles = plan->phase_cycles;
	args.k5const = plan->d_consts5;

	EventPair ev;
	if (plan->timing) {
		if (!plan->pool.empty()) {
			ev = plan->pool.back();
			plan->pool.pop_back();
		} else {
			if ((e = hipEventCreate(&ev.start)) != hipSuccess) return fail_hip(e, "hipEventCreate");
			if ((e = hipEventCreate(&ev.stop)) != hipSuccess) return fail_hip(e, "hipEventCreate");
		}
		if ((e = hipEventRecord(ev.start, stream)) != hipSuccess) return fail_hip(e, "hipEventRecord");
	}
	e = model5 ? gvtm::launch_synth5(args, batch, rows, stream)
	           : gvtm::launch_synth(args, batch, plan->precision, rows, stream);
	if (plan->timing) {
		(void) hipEventRecord(ev.stop, stream);
		plan->pending.push_back(ev);
	}
	if (e != hipSuccess) return fail_hip(e, "vtm_synth_kernel launch");
	return GVTM_OK;
}

} // namespace

extern "C" {

int gvtm_synthesize_batch_device(gvtm_plan* plan, const float* d_params, const int32_t* d_frame_counts,
		size_t batch, size_t max_frames, float* d_audio, size_t audio_stride,
		int64_t* d_out_counts, float* d_maxabs, void* hip_stream)
{
	return launch_batch(plan, d_params, d_frame_counts, batch, max_frames, d_audio, audio_stride, d_out_counts, d_maxabs, hip_stream, nullptr);
}

int gvtm_synthesize_events_device(gvtm_plan* plan, const gvtm_track_config* config, const gvtm_event* d_events,
		const int64_t* d_event_offsets, size_t batch, size_t max_frames, float* d_audio, size_t audio_stride,
		int32_t* d_frame_counts, int64_t* d_out_counts, float* d_maxabs, gvtm_drift_state* d_drift, void* hip_stream)
{
	if (!plan || !config) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan or config");
	if (plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_NO_DEVICE, "design-only plan (GVTM_DEVICE_NONE): there is no CPU synthesis path");
	if (batch == 0) return GVTM_OK;
	if (!d_events || !d_event_offsets) return fail(GVTM_ERR_INVALID_ARGUMENT, "null events or event_offsets");
	gvtm::TrackConstants tk{};
	const char* why = gvtm::design_tracks(*config, tk);
	if (why[0]) return fail(GVTM_ERR_INVALID_ARGUMENT, why);
	if (static_cast<double>(tk.control_period) * plan->design.control_rate != 1000.0) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, "control_period_ms of the track configuration and the plan's control rate disagree");
	}
	// Two launches on the caller's stream with a frame buffer of the plan's in between.  (Walking the event lists inside the
	// synthesis kernel's interpolation wavefront was built and measured: bit-identical, no frame buffer, and 17.2 -> 33.1 ms
	// per 4096 x 80 events -- an event boundary is a round trip to memory in the middle of a tick, three times over because
	// the parameter groups run at different lags -- and 125 ms with the next events prefetched into registers, which that
	// wavefront does not have to spare.  DESIGN.md 6b.)
	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	if ((e = plan->s_params.ensure(sizeof(float) * batch * std::max<size_t>(max_frames, 1) * GVTM_N_PARAM)) != hipSuccess) return fail_hip(e, "hipMalloc frames");
	if ((e = plan->s_frames.ensure(sizeof(int32_t) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc frame counts");
	int32_t* const counts = d_frame_counts ? d_frame_counts : static_cast<int32_t*>(plan->s_frames.ptr);
	int rc = gvtm_generate_tracks_device(plan->device, config, d_events, d_event_offsets, batch, max_frames, static_cast<float*>(plan->s_params.ptr), counts,
			d_drift, hip_stream);
	if (rc != GVTM_OK) return rc;
	return launch_batch(plan, static_cast<const float*>(plan->s_params.ptr), counts, batch, max_frames, d_audio, audio_stride, d_out_counts, d_maxabs,
			hip_stream, nullptr);
}

} // extern "C"

namespace {

// The host-buffer entries: frames in host memory -> samples in host memory, float32 (unscaled outputBuffer() samples) or
// int16 (scaled by 0.95 / max|x| and rounded as WAVEFileWriter::writeSample does, Controller.cpp:315-340,
// WAVEFileWriter.cpp:122-125 -- half the bytes over PCIe, which is what bounds this entry).
struct HostJob {
	const float* params;
	const int32_t* frame_counts;
	size_t batch, max_frames;
	float* audio;      // float32 output [batch][stride], or null
	int16_t* pcm;      // int16 output [batch][stride], or null
	size_t stride;
	int64_t* out_counts;
	float* maxabs;
	float* scales;     // pcm only: the scale applied to each utterance, or null
};

// A batch of at least two machine-fulls goes in slices of one machine-full each (rows x compute units utterances: every
// compute unit busy, in the shape the whole batch would use), three streams deep:
//     H2D frames(i + 1)  ||  kernel(i) [+ scale -> int16(i)]  ||  D2H samples(i - 1)
// With page-locked host buffers (gvtm_host_alloc) all three really overlap; with pageable ones the runtime stages the
// copies itself and the call still returns the same bytes.
int host_pipeline(gvtm_plan* plan, const HostJob& j)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	if (plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_NO_DEVICE, "design-only plan (GVTM_DEVICE_NONE): there is no CPU synthesis path");
	const size_t batch = j.batch, max_frames = j.max_frames, audio_stride = j.stride;
	if (batch == 0) return GVTM_OK;
	if (!j.audio && !j.pcm) return fail(GVTM_ERR_INVALID_ARGUMENT, "null output buffer");
	if (max_frames > 0 && !j.params) return fail(GVTM_ERR_INVALID_ARGUMENT, "null params with max_frames > 0");
	if (audio_stride < gvtm_output_count(plan, max_frames)) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, "audio_stride smaller than gvtm_output_count(plan, max_frames)");
	}
	// A frame count outside [0, max_frames] fails THAT utterance (out_counts[b] = -1, no samples); the others are
	// synthesized.  The device sees it as an empty utterance.
	std::vector<int32_t> sane;
	std::vector<size_t> bad;
	try {
		if (j.frame_counts) {
			for (size_t b = 0; b < batch; ++b) {
				if (j.frame_counts[b] < 0 || static_cast<size_t>(j.frame_counts[b]) > max_frames) {
					if (sane.empty()) sane.assign(j.frame_counts, j.frame_counts + batch);
					sane[b] = 0;
					bad.push_back(b);
				}
			}
		}
	} catch (const std::bad_alloc&) {
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	}
	const int32_t* const counts_in = sane.empty() ? j.frame_counts : sane.data();
	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	const size_t row_in = max_frames * GVTM_N_PARAM;
	const size_t pbytes = sizeof(float) * batch * row_in;
	const size_t abytes = sizeof(float) * batch * audio_stride;
	const size_t obytes = sizeof(int16_t) * batch * audio_stride;
	if ((e = plan->s_params.ensure(pbytes ? pbytes : 16)) != hipSuccess) return fail_hip(e, "hipMalloc params");
	if ((e = plan->s_audio.ensure(abytes ? abytes : 16)) != hipSuccess) return fail_hip(e, "hipMalloc audio");
	if ((e = plan->s_counts.ensure(sizeof(int64_t) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc counts");
	if ((e = plan->s_maxabs.ensure(sizeof(float) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc maxabs");
	if (counts_in && (e = plan->s_frames.ensure(sizeof(int32_t) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc frames");
	if (j.pcm && (e = plan->s_pcm.ensure(obytes ? obytes : 16)) != hipSuccess) return fail_hip(e, "hipMalloc pcm");
	if (j.pcm && (e = plan->s_scales.ensure(sizeof(float) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc scales");
	if (!plan->h2d_stream && (e = hipStreamCreateWithFlags(&plan->h2d_stream, hipStreamNonBlocking)) != hipSuccess) return fail_hip(e, "hipStreamCreate");
	if (!plan->compute_stream && (e = hipStreamCreateWithFlags(&plan->compute_stream, hipStreamNonBlocking)) != hipSuccess) return fail_hip(e, "hipStreamCreate");
	if (!plan->copy_stream && (e = hipStreamCreateWithFlags(&plan->copy_stream, hipStreamNonBlocking)) != hipSuccess) return fail_hip(e, "hipStreamCreate");

	float* const d_params = static_cast<float*>(plan->s_params.ptr);
	const int32_t* const d_frames = counts_in ? static_cast<const int32_t*>(plan->s_frames.ptr) : nullptr;
	float* const d_audio = static_cast<float*>(plan->s_audio.ptr);
	int16_t* const d_pcm = j.pcm ? static_cast<int16_t*>(plan->s_pcm.ptr) : nullptr;
	float* const d_scales = j.pcm ? static_cast<float*>(plan->s_scales.ptr) : nullptr;
	int64_t* const d_counts = static_cast<int64_t*>(plan->s_counts.ptr);
	float* const d_maxabs = static_cast<float*>(plan->s_maxabs.ptr);
	if (counts_in && (e = hipMemcpy(plan->s_frames.ptr, counts_in, sizeof(int32_t) * batch, hipMemcpyHostToDevice)) != hipSuccess) {
		return fail_hip(e, "H2D frame_counts");
	}
	// rows come back zero beyond their sample count (the staging buffers are reused between calls)
	const bool ragged = counts_in != nullptr || audio_stride > gvtm_output_count(plan, max_frames);

	// the shape of the whole batch, and how many utterances fill the machine once in it
	const bool model5 = plan->design.model5;
	const int rows_all = model5 ? plan->rows5_for(batch) : gvtm::synth_rows(plan->precision, batch, plan->rows, plan->design.k.section_delay);
	const size_t machine = static_cast<size_t>(rows_all) * static_cast<size_t>(plan->compute_units > 0 ? plan->compute_units : 256);
	const size_t slice = batch >= 2 * machine ? machine : batch;
	const size_t n_slices = (batch + slice - 1) / slice;
	while (plan->slice_done.size() < 2 * n_slices) {
		hipEvent_t ev = nullptr;
		if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail_hip(e, "hipEventCreate");
		plan->slice_done.push_back(ev);
	}
	auto drain = [&]() {
		(void) hipStreamSynchronize(plan->h2d_stream);
		(void) hipStreamSynchronize(plan->compute_stream);
		(void) hipStreamSynchronize(plan->copy_stream);
	};
	const int saved_rows = plan->rows;
	if (plan->rows == 0) plan->rows = rows_all; // every slice in the whole batch's shape
	int rc = GVTM_OK;
	for (size_t i = 0; i < n_slices && rc == GVTM_OK; ++i) {
		const size_t lo = i * slice, n = std::min(slice, batch - lo);
		hipEvent_t in_done = plan->slice_done[2 * i], out_ready = plan->slice_done[2 * i + 1];
		if (pbytes && (e = hipMemcpyAsync(d_params + lo * row_in, j.params + lo * row_in, sizeof(float) * n * row_in, hipMemcpyHostToDevice,
				plan->h2d_stream)) != hipSuccess) { rc = fail_hip(e, "H2D params"); break; }
		if ((e = hipEventRecord(in_done, plan->h2d_stream)) != hipSuccess) { rc = fail_hip(e, "hipEventRecord"); break; }
		if ((e = hipStreamWaitEvent(plan->compute_stream, in_done, 0)) != hipSuccess) { rc = fail_hip(e, "hipStreamWaitEvent"); break; }
		if (ragged && !j.pcm && (e = hipMemsetAsync(d_audio + lo * audio_stride, 0, sizeof(float) * n * audio_stride, plan->compute_stream)) != hipSuccess) {
			rc = fail_hip(e, "hipMemsetAsync"); break;
		}
		if (ragged && j.pcm && (e = hipMemsetAsync(d_pcm + lo * audio_stride, 0, sizeof(int16_t) * n * audio_stride, plan->compute_stream)) != hipSuccess) {
			rc = fail_hip(e, "hipMemsetAsync"); break;
		}
		rc = gvtm_synthesize_batch_device(plan, d_params + lo * row_in, d_frames ? d_frames + lo : nullptr, n, max_frames,
				d_audio + lo * audio_stride, audio_stride, d_counts + lo, d_maxabs + lo, plan->compute_stream);
		if (rc != GVTM_OK) break;
		if (j.pcm) {
			// (normalize takes at most 65535 utterances per launch: a slice is far below that unless the batch is one slice)
			for (size_t q = 0; q < n && rc == GVTM_OK; q += 32768) {
				const size_t m = std::min<size_t>(32768, n - q);
				rc = gvtm_normalize_batch_device(plan, d_audio + (lo + q) * audio_stride, m, audio_stride, d_counts + lo + q, d_maxabs + lo + q, nullptr,
						d_pcm + (lo + q) * audio_stride, d_scales + lo + q, plan->compute_stream);
			}
			if (rc != GVTM_OK) break;
		}
		if ((e = hipEventRecord(out_ready, plan->compute_stream)) != hipSuccess) { rc = fail_hip(e, "hipEventRecord"); break; }
	}
	plan->rows = saved_rows;
	if (rc != GVTM_OK) {
		drain();
		return rc;
	}
	// (second loop: with pageable host memory a device-to-host copy blocks the calling thread until its slice is done,
	// so every kernel is queued before the first of them)
	for (size_t i = 0; i < n_slices; ++i) {
		const size_t lo = i * slice, n = std::min(slice, batch - lo);
		if ((e = hipStreamWaitEvent(plan->copy_stream, plan->slice_done[2 * i + 1], 0)) != hipSuccess) { drain(); return fail_hip(e, "hipStreamWaitEvent"); }
		if (j.pcm) e = hipMemcpyAsync(j.pcm + lo * audio_stride, d_pcm + lo * audio_stride, sizeof(int16_t) * n * audio_stride, hipMemcpyDeviceToHost, plan->copy_stream);
		else e = hipMemcpyAsync(j.audio + lo * audio_stride, d_audio + lo * audio_stride, sizeof(float) * n * audio_stride, hipMemcpyDeviceToHost, plan->copy_stream);
		if (e != hipSuccess) { drain(); return fail_hip(e, "D2H samples"); }
	}
	if ((e = hipStreamSynchronize(plan->copy_stream)) != hipSuccess) { drain(); return fail_hip(e, "vtm_synth_kernel execution / D2H samples"); }
	if ((e = hipStreamSynchronize(plan->compute_stream)) != hipSuccess) return fail_hip(e, "vtm_synth_kernel execution");
	if ((e = hipStreamSynchronize(plan->h2d_stream)) != hipSuccess) return fail_hip(e, "H2D params");
	// (everything this plan has launched so far is complete: superseded noise tables can go)
	for (void* q : plan->noise_retired) (void) hipFree(q);
	plan->noise_retired.clear();
	if (j.out_counts && (e = hipMemcpy(j.out_counts, d_counts, sizeof(int64_t) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return fail_hip(e, "D2H counts");
	if (j.maxabs && (e = hipMemcpy(j.maxabs, d_maxabs, sizeof(float) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return fail_hip(e, "D2H maxabs");
	if (j.scales && d_scales && (e = hipMemcpy(j.scales, d_scales, sizeof(float) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return fail_hip(e, "D2H scales");
	for (size_t b : bad) {
		if (j.out_counts) j.out_counts[b] = -1;
		if (j.maxabs) j.maxabs[b] = 0.0f;
		if (j.scales) j.scales[b] = 0.0f;
		if (j.audio) std::fill(j.audio + b * audio_stride, j.audio + (b + 1) * audio_stride, 0.0f);
		if (j.pcm) std::fill(j.pcm + b * audio_stride, j.pcm + (b + 1) * audio_stride, int16_t(0));
	}
	if (!bad.empty()) g_last_error = "frame_counts entry outside [0, max_frames]: those utterances have out_counts = -1";
	return GVTM_OK;
}

} // namespace

extern "C" {

int gvtm_synthesize_batch_host(gvtm_plan* plan, const float* params, const int32_t* frame_counts,
		size_t batch, size_t max_frames, float* audio, size_t audio_stride,
		int64_t* out_counts, float* maxabs)
{
	if (batch != 0 && !audio) return fail(GVTM_ERR_INVALID_ARGUMENT, "null audio buffer");
	return host_pipeline(plan, HostJob{params, frame_counts, batch, max_frames, audio, nullptr, audio_stride, out_counts, maxabs, nullptr});
}

int gvtm_synthesize_batch_host_pcm16(gvtm_plan* plan, const float* params, const int32_t* frame_counts,
		size_t batch, size_t max_frames, int16_t* pcm, size_t pcm_stride,
		int64_t* out_counts, float* maxabs, float* scales)
{
	if (batch != 0 && !pcm) return fail(GVTM_ERR_INVALID_ARGUMENT, "null pcm buffer");
	return host_pipeline(plan, HostJob{params, frame_counts, batch, max_frames, nullptr, pcm, pcm_stride, out_counts, maxabs, scales});
}

int gvtm_host_alloc(size_t bytes, void** ptr_out)
{
	if (!ptr_out) return fail(GVTM_ERR_INVALID_ARGUMENT, "null ptr_out");
	*ptr_out = nullptr;
	if (bytes == 0) return GVTM_OK;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(GVTM_ERR_NO_DEVICE, "no HIP device available: page-locked memory comes from the HIP runtime");
	const hipError_t e = hipHostMalloc(ptr_out, bytes, hipHostMallocPortable);
	if (e != hipSuccess) { *ptr_out = nullptr; return e == hipErrorOutOfMemory ? fail(GVTM_ERR_OUT_OF_MEMORY, "hipHostMalloc: out of memory") : fail_hip(e, "hipHostMalloc"); }
	return GVTM_OK;
}

void gvtm_host_free(void* ptr)
{
	if (ptr) (void) hipHostFree(ptr);
}

/* ---------------------------------------------------------------------------------------------
 * Streams: the stateful form of the path (include/gama_vtm.h, "Streams").
 */

} // extern "C"

struct gvtm_stream {
	gvtm_plan* plan = nullptr;
	size_t batch = 0;
	size_t state_stride = 0;
	int xr = 0;
	unsigned granule_frames = 1;          // pushes are synthesized in multiples of this many frames (12 internal steps)
	DeviceBuffer d_state, d_params, d_frames, d_audio, d_counts, d_maxabs;
	std::vector<std::vector<float>> held; // per utterance: frames pushed but not yet synthesized (the last one is the look-ahead)
	std::vector<uint64_t> steps_done;     // per utterance: internal steps synthesized
	std::vector<float> staging;
	std::vector<int32_t> counts;
	bool finished = false;
};

namespace {

unsigned gcd_u(unsigned a, unsigned b)
{
	while (b) { const unsigned t = a % b; a = b; b = t; }
	return a;
}

int stream_upload_fresh_state(gvtm_stream* s)
{
	std::vector<unsigned char> init(s->state_stride * s->batch, 0);
	const bool model5 = s->plan->design.model5;
	for (size_t b = 0; b < s->batch; ++b) {
		gvtm::StreamHeader h{};
		h.seed = 0.7892347; // NoiseSource::reset (vtm/NoiseSource.h:32-34)
		std::memcpy(init.data() + b * s->state_stride, &h, sizeof(h));
		if (model5) {
			// VocalTractModel5::reset (vtm/VocalTractModel5.h:423-453): RosenbergBGlottalSource::reset leaves t2 at the end of the
			// longest falling phase, the noise source starts from its seed, everything else is zero
			const gvtm::Model5Constants& k5 = s->plan->design.k5;
			double sc[gvtm::kStream5Scalars] = {};
			sc[gvtm::kS5Scan + 1] = k5.rb_t1 + k5.rb_tn_max;
			sc[gvtm::kS5Scan + 2] = 0.7892347;
			std::memcpy(init.data() + b * s->state_stride + gvtm::Stream5Layout::scalars(), sc, sizeof(sc));
		}
	}
	hipError_t e = hipMemcpy(s->d_state.ptr, init.data(), init.size(), hipMemcpyHostToDevice);
	if (e != hipSuccess) return fail_hip(e, "H2D stream state");
	return GVTM_OK;
}

uint64_t outputs_before(const gvtm::DeviceConstants& k, uint64_t steps)
{
	return ((steps << 16) + k.time_inc - 1) / k.time_inc;
}

// One launch on behalf of the stream: utterance b synthesizes n_frames[b] frames from the front of held[b].
int stream_launch(gvtm_stream* s, const std::vector<size_t>& n_frames, bool final, float* audio, size_t audio_stride, int64_t* out_counts,
		float* maxabs, bool* launched = nullptr)
{
	gvtm_plan* plan = s->plan;
	const gvtm::DeviceConstants& k = plan->design.k;
	const size_t batch = s->batch;
	size_t rows_max = 0;
	bool lockstep = true, any = final;
	for (size_t b = 0; b < batch; ++b) {
		const size_t rows = n_frames[b] + (final ? 0 : 1); // a push carries the look-ahead frame behind its last one
		rows_max = std::max(rows_max, n_frames[b] ? rows : size_t(0));
		if (n_frames[b] != n_frames[0] || s->steps_done[b] != s->steps_done[0]) lockstep = false;
		if (n_frames[b]) any = true;
	}
	// exact sample counts, known before the launch
	size_t need = 0;
	std::vector<int64_t> want(batch, 0);
	for (size_t b = 0; b < batch; ++b) {
		const uint64_t after = s->steps_done[b] + static_cast<uint64_t>(n_frames[b]) * k.control_steps;
		if (after + 4096ull >= (1ull << 31)) return fail(GVTM_ERR_INVALID_ARGUMENT, "a stream holds at most 2^31 internal steps between resets");
		const uint64_t k0 = outputs_before(k, s->steps_done[b]);
		const uint64_t k1 = final ? gvtm::src_output_count(k.time_inc, k.pad, k.upsampling, after) : outputs_before(k, after);
		want[b] = static_cast<int64_t>(k1 - k0);
		need = std::max(need, static_cast<size_t>(want[b]));
	}
	if (need > audio_stride) return fail(GVTM_ERR_INVALID_ARGUMENT, "audio_stride smaller than this call produces (gvtm_stream_capacity)");
	if (need > 0 && !audio) return fail(GVTM_ERR_INVALID_ARGUMENT, "null audio buffer");
	if (!any) {
		if (out_counts) std::fill(out_counts, out_counts + batch, int64_t(0));
		return GVTM_OK;
	}
	if (rows_max == 0) rows_max = 1;
	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	try {
		s->staging.assign(batch * rows_max * GVTM_N_PARAM, 0.0f);
		s->counts.assign(batch, 0);
	} catch (const std::bad_alloc&) {
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	}
	for (size_t b = 0; b < batch; ++b) {
		if (!n_frames[b]) continue;
		const size_t rows = n_frames[b] + (final ? 0 : 1);
		std::memcpy(s->staging.data() + b * rows_max * GVTM_N_PARAM, s->held[b].data(), sizeof(float) * rows * GVTM_N_PARAM);
		s->counts[b] = static_cast<int32_t>(n_frames[b]);
	}
	const size_t pbytes = sizeof(float) * s->staging.size();
	const size_t abytes = sizeof(float) * batch * std::max<size_t>(audio_stride, 1);
	if ((e = s->d_params.ensure(pbytes)) != hipSuccess) return fail_hip(e, "hipMalloc params");
	if ((e = s->d_frames.ensure(sizeof(int32_t) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc frames");
	if ((e = s->d_audio.ensure(abytes)) != hipSuccess) return fail_hip(e, "hipMalloc audio");
	if ((e = s->d_counts.ensure(sizeof(int64_t) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc counts");
	if ((e = s->d_maxabs.ensure(sizeof(float) * batch)) != hipSuccess) return fail_hip(e, "hipMalloc maxabs");
	if ((e = hipMemcpy(s->d_params.ptr, s->staging.data(), pbytes, hipMemcpyHostToDevice)) != hipSuccess) return fail_hip(e, "H2D params");
	if ((e = hipMemcpy(s->d_frames.ptr, s->counts.data(), sizeof(int32_t) * batch, hipMemcpyHostToDevice)) != hipSuccess) return fail_hip(e, "H2D frame counts");
	if ((e = hipMemsetAsync(s->d_audio.ptr, 0, abytes, nullptr)) != hipSuccess) return fail_hip(e, "hipMemsetAsync");
	StreamLaunch sl{static_cast<unsigned char*>(s->d_state.ptr), s->state_stride, final ? gvtm::kStreamFinish : gvtm::kStreamPush, s->xr, lockstep ? 0 : 1};
	const int rc = launch_batch(plan, static_cast<const float*>(s->d_params.ptr), static_cast<const int32_t*>(s->d_frames.ptr), batch, rows_max,
			static_cast<float*>(s->d_audio.ptr), audio_stride, static_cast<int64_t*>(s->d_counts.ptr), static_cast<float*>(s->d_maxabs.ptr), nullptr, &sl);
	if (rc != GVTM_OK) return rc;
	if (launched) *launched = true;
	if ((e = hipDeviceSynchronize()) != hipSuccess) return fail_hip(e, "vtm_synth_kernel execution");
	std::vector<int64_t> got(batch, 0);
	if ((e = hipMemcpy(got.data(), s->d_counts.ptr, sizeof(int64_t) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return fail_hip(e, "D2H counts");
	for (size_t b = 0; b < batch; ++b) {
		if (got[b] != want[b]) return fail(GVTM_ERR_HIP, "internal error: the device's sample count differs from the host's");
	}
	if (audio && audio_stride && (e = hipMemcpy(audio, s->d_audio.ptr, sizeof(float) * batch * audio_stride, hipMemcpyDeviceToHost)) != hipSuccess) {
		return fail_hip(e, "D2H audio");
	}
	if (out_counts) std::copy(got.begin(), got.end(), out_counts);
	if (maxabs && (e = hipMemcpy(maxabs, s->d_maxabs.ptr, sizeof(float) * batch, hipMemcpyDeviceToHost)) != hipSuccess) return fail_hip(e, "D2H maxabs");
	for (size_t b = 0; b < batch; ++b) {
		s->steps_done[b] += static_cast<uint64_t>(n_frames[b]) * k.control_steps;
		s->held[b].erase(s->held[b].begin(), s->held[b].begin() + static_cast<std::ptrdiff_t>(n_frames[b] * GVTM_N_PARAM));
	}
	return GVTM_OK;
}

} // namespace

extern "C" {

int gvtm_stream_create(gvtm_plan* plan, size_t batch, gvtm_stream** stream_out)
{
	if (!plan || !stream_out || batch == 0) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan / stream_out or empty batch");
	*stream_out = nullptr;
	if (plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_NO_DEVICE, "design-only plan (GVTM_DEVICE_NONE): there is no CPU synthesis path");
	try {
		std::unique_ptr<gvtm_stream> s(new gvtm_stream);
		s->plan = plan;
		s->batch = batch;
		const gvtm::DeviceConstants& k = plan->design.k;
		if (plan->design.model5) {
			// reference model 5: its own state block (vtm_kernels.hpp: Stream5Layout); the serial wavefronts work in blocks
			// of four steps (vtm_kernel_m5.inc)
			s->xr = 0;
			s->state_stride = gvtm::Stream5Layout::bytes();
			s->granule_frames = 4u / gcd_u(k.control_steps, 4u);
		} else {
			s->xr = gvtm::synth_ring_length(k, plan->precision, 1);
			s->state_stride = gvtm::stream_state_bytes(k, plan->precision, s->xr);
			// the serial wavefronts work in blocks of 2, 4 and 4 or 6 steps (vtm_kernel_v2.inc): their states are exact at
			// multiples of 12 steps, so a push synthesizes a multiple of 12 / gcd(control_steps, 12) frames and keeps the rest
			s->granule_frames = 12u / gcd_u(k.control_steps, 12u);
		}
		s->held.resize(batch);
		s->steps_done.assign(batch, 0);
		DeviceScope scope(plan->device);
		hipError_t e = scope.status();
		if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
		if ((e = s->d_state.ensure(s->state_stride * batch)) != hipSuccess) return fail_hip(e, "hipMalloc stream state");
		const int rc = stream_upload_fresh_state(s.get());
		if (rc != GVTM_OK) { s->d_state.release(); return rc; }
		*stream_out = s.release();
		return GVTM_OK;
	} catch (const std::bad_alloc&) {
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	}
}

void gvtm_stream_destroy(gvtm_stream* s)
{
	if (!s) return;
	{
		DeviceScope scope(s->plan->device);
		s->d_state.release(); s->d_params.release(); s->d_frames.release(); s->d_audio.release(); s->d_counts.release(); s->d_maxabs.release();
	}
	delete s;
}

int gvtm_stream_reset(gvtm_stream* s)
{
	if (!s) return fail(GVTM_ERR_INVALID_ARGUMENT, "null stream");
	DeviceScope scope(s->plan->device);
	if (scope.status() != hipSuccess) return fail_hip(scope.status(), "hipSetDevice");
	for (auto& h : s->held) h.clear();
	std::fill(s->steps_done.begin(), s->steps_done.end(), uint64_t(0));
	s->finished = false;
	return stream_upload_fresh_state(s);
}

size_t gvtm_stream_capacity(const gvtm_stream* s, size_t max_new_frames)
{
	if (!s) return static_cast<size_t>(-1);
	const gvtm::DeviceConstants& k = s->plan->design.k;
	// at most the new frames plus what a push can have kept (granule_frames frames), flushed, with the overrun's lap
	const uint64_t steps = static_cast<uint64_t>(max_new_frames + s->granule_frames) * k.control_steps;
	return static_cast<size_t>(gvtm::src_output_capacity(k.time_inc, k.pad, k.upsampling, steps) + 1);
}

int gvtm_stream_push(gvtm_stream* s, const float* params, const int32_t* frame_counts, size_t max_frames,
		float* audio, size_t audio_stride, int64_t* out_counts)
{
	if (!s) return fail(GVTM_ERR_INVALID_ARGUMENT, "null stream");
	if (s->finished) return fail(GVTM_ERR_INVALID_ARGUMENT, "the stream has been finished: gvtm_stream_reset() starts the next utterances");
	if (max_frames > 0 && !params) return fail(GVTM_ERR_INVALID_ARGUMENT, "null params with max_frames > 0");
	if (frame_counts) {
		for (size_t b = 0; b < s->batch; ++b) {
			if (frame_counts[b] < 0 || static_cast<size_t>(frame_counts[b]) > max_frames) return fail(GVTM_ERR_INVALID_ARGUMENT, "frame_counts entry outside [0, max_frames]");
		}
	}
	// a call that fails (a stride too small, a null buffer, an allocation) leaves the stream as it found it: the frames it
	// appended are taken back, so that the corrected call does not synthesize them twice
	std::vector<size_t> held_before(s->batch, 0);
	for (size_t b = 0; b < s->batch; ++b) held_before[b] = s->held[b].size();
	auto roll_back = [&]() {
		for (size_t b = 0; b < s->batch; ++b) {
			if (s->held[b].size() > held_before[b]) s->held[b].resize(held_before[b]);
		}
	};
	try {
		std::vector<size_t> n(s->batch, 0);
		for (size_t b = 0; b < s->batch; ++b) {
			const size_t add = frame_counts ? static_cast<size_t>(frame_counts[b]) : max_frames;
			const float* src = params + b * max_frames * GVTM_N_PARAM;
			s->held[b].insert(s->held[b].end(), src, src + add * GVTM_N_PARAM);
			const size_t have = s->held[b].size() / GVTM_N_PARAM;
			// the last frame held is the look-ahead of the one before it (Controller.cpp:297-300 interpolates towards the NEXT frame)
			n[b] = have > 0 ? ((have - 1) / s->granule_frames) * s->granule_frames : 0;
		}
		bool launched = false;
		const int rc = stream_launch(s, n, false, audio, audio_stride, out_counts, nullptr, &launched);
		if (rc != GVTM_OK && !launched) roll_back(); // (after the launch the device state has moved on: gvtm_stream_reset is the way out)
		return rc;
	} catch (const std::bad_alloc&) {
		roll_back();
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	}
}

int gvtm_stream_finish(gvtm_stream* s, float* audio, size_t audio_stride, int64_t* out_counts, float* maxabs)
{
	if (!s) return fail(GVTM_ERR_INVALID_ARGUMENT, "null stream");
	if (s->finished) return fail(GVTM_ERR_INVALID_ARGUMENT, "the stream has already been finished");
	try {
		std::vector<size_t> n(s->batch, 0);
		for (size_t b = 0; b < s->batch; ++b) n[b] = s->held[b].size() / GVTM_N_PARAM; // the last frame stands for its own successor (Controller.cpp:283)
		const int rc = stream_launch(s, n, true, audio, audio_stride, out_counts, maxabs);
		if (rc == GVTM_OK) s->finished = true;
		return rc;
	} catch (const std::bad_alloc&) {
		return fail(GVTM_ERR_OUT_OF_MEMORY, "host allocation failed");
	}
}

int gvtm_normalize_batch_device(gvtm_plan* plan, const float* d_audio, size_t batch, size_t audio_stride,
		const int64_t* d_counts, const float* d_maxabs, float* d_out_f32, int16_t* d_out_i16,
		float* d_scales, void* hip_stream)
{
	if (!plan) return fail(GVTM_ERR_INVALID_ARGUMENT, "null plan");
	if (plan->device == GVTM_DEVICE_NONE) return fail(GVTM_ERR_NO_DEVICE, "design-only plan (GVTM_DEVICE_NONE)");
	if (batch == 0 || audio_stride == 0) return GVTM_OK;
	if (!d_audio || !d_maxabs) return fail(GVTM_ERR_INVALID_ARGUMENT, "null audio or maxabs");
	if ((d_out_f32 == nullptr) == (d_out_i16 == nullptr)) {
		return fail(GVTM_ERR_INVALID_ARGUMENT, "exactly one of d_out_f32 / d_out_i16 must be given");
	}
	if (batch > 65535) return fail(GVTM_ERR_INVALID_ARGUMENT, "normalize: batch above 65535 per call");
	DeviceScope scope(plan->device);
	hipError_t e = scope.status();
	if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
	gvtm::NormalizeArgs args;
	args.audio = d_audio;
	args.counts = d_counts;
	args.maxabs = d_maxabs;
	args.out_f32 = d_out_f32;
	args.out_i16 = d_out_i16;
	args.scales = d_scales;
	args.audio_stride = audio_stride;
	e = gvtm::launch_normalize(args, batch, static_cast<hipStream_t>(hip_stream));
	if (e != hipSuccess) return fail_hip(e, "vtm_normalize_kernel launch");
	return GVTM_OK;
}

} // extern "C"
