// Host-side design of everything that is fixed for a voice configuration:
// derived rates, filter constants and the fp64 tables the device kernels stage
// in LDS.  All of it is computed in fp64 on the host exactly once per plan
// (design tables must be fp64 even for the mixed-precision path).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gama_vtm.h"

namespace gvtm {

constexpr int kWavetableLength = 512;   // WavetableGlottalSource.h:91
constexpr int kSrcZeroCrossings = 13;   // SampleRateConverter.h:49
constexpr int kSrcPhases = 256;         // L_RANGE, SampleRateConverter.h:46
constexpr int kSrcFilterLength = kSrcZeroCrossings * kSrcPhases; // 3328
constexpr int kSrcRing = 1024;          // BUFFER_SIZE, SampleRateConverter.h:44
constexpr int kMaxFirTaps = 64;
constexpr int kMaxSectionDelay = 4;
constexpr int kMaxPad = 96;

// Everything the kernels need besides the tables; plain data, copied to the device by value.
struct DeviceConstants {
	// rates and driver loop
	int sample_rate;          // internal rate
	unsigned control_steps;   // steps per frame
	float interp_coef;        // 1.0f / control_steps (float32, Controller.cpp:287)
	int section_delay;
	int layout;               // gvtm_tube_layout
	// sources
	int waveform;             // 0 pulse, 1 sine
	int modulation;
	unsigned table_div1, table_div2;
	double tn_delta;          // 0 => static wavetable
	double basic_increment;   // 512 / fs
	double breathiness;       // breathiness / 100
	double crossmix_factor;   // 1 / amp60(mix_offset)
	int fir_taps;
	// tube
	double damping;
	double radius_coef[8];
	double aperture_radius2;  // apertureRadius^2
	double nasal_r2_sq;       // nasalRadius[N2]^2 (first nasal junction)
	double nasal_k[6];        // fixed nasal coefficients; [0] unused (time varying)
	double mouth_b0_refl, mouth_a1_refl, mouth_a_rad; // ReflectionFilter / RadiationFilter constants
	double nose_b0_refl, nose_a1_refl, nose_a_rad;
	double throat_b0, throat_a1, throat_gain;
	double bp_T;              // 1 / fs for BandpassFilter::update
	// sample-rate converter
	int upsampling;
	int pad;
	unsigned time_inc;        // timeRegisterIncrement_
	unsigned phase_inc;       // phaseIncrement_ (down-sampling)
	double src_ratio;         // sampleRateRatio_
};

// VocalTractModel5 only: what initializeSynthesizer (vtm/VocalTractModel5.h:455-521) derives besides the numbers
// shared with the other models (rates, driver loop, radius coefficients, damping, resampler in DeviceConstants)
struct Model5Constants {
	double sample_rate;        // sampleRate_ (TFloat, not truncated)
	double output_rate;
	int bypass, constant_mouth;
	double rb_t1, rb_tn_min, rb_tn_max;      // RosenbergBGlottalSource (vtm/RosenbergBGlottalSource.h:66-96)
	double gn_b0, gn_a1;                     // glottal-noise Butterworth1 (vtm/Butterworth1LowpassFilter.h:63-76)
	double gp_b0, gp_a1;                     // glottal-pulse Butterworth1
	double fn_b0, fn_b1, fn_a1, fn_a2;       // frication-noise Butterworth2 (vtm/Butterworth2LowpassFilter.h:88-107)
	double frication_factor;
	double min_loss, max_loss;               // min/max_glottal_loss / 100
	double nasal_r1_sq;                      // nasalRadius[NR2]^2: right side of the first nasal junction
	double nasal_k[6];                       // [1..5]: fixed nasal junctions NJ2..NJ6; [0] unused (time varying)
	double period;                           // PoleZeroRadiationImpedance::samplePeriod_
	double nose_c[6];                        // cT1, cT2, cT3, cR1, cR2, cR3 at the nose (fixed radius)
	double mouth_c[6];                       // the same at the mouth when its radius is constant
};

struct Design {
	bool model5 = false;
	gvtm5_config config5{};
	Model5Constants k5{};
	gvtm_config config;
	double control_rate;
	DeviceConstants k;
	std::vector<double> fir;        // fir_taps
	std::vector<double> src_h;      // 3328
	std::vector<double> src_dh;     // 3328
	std::vector<double> wavetable;  // 512
	// GVTM_PRECISION_F32: the tables as designed in float (the double vectors above hold the same values widened)
	bool f32 = false;
	std::vector<float> fir_f, src_h_f, src_dh_f, wavetable_f;
};

// The noise source's low-passed samples for internal steps [0, n) (NoiseSource.h:40-44 with its fixed seed,
// NoiseFilter.h:63-68): every utterance of every batch draws the same sequence after reset(), so a plan tabulates it
// once; `out` holds n floats (as_float: the sum formed in float as NoiseFilter<float> does) or n doubles.
void design_noise_table(size_t n, bool as_float, void* out);

// Returns "" on success, otherwise a description of the offending value.
std::string design_plan(const gvtm_config& cfg, double control_rate, Design& out);

std::string design_plan5(const gvtm5_config& cfg, double control_rate, Design& out);

// PoleZeroRadiationImpedance::update (vtm/PoleZeroRadiationImpedance.h:139-177): radius in metres -> cT1..3, cR1..3
void radiation_impedance(double radius, double period, double out[6]);

// Util::amplitude60dB (vtm/VTMUtil.h:48-67)
double amplitude_60db(double db);

#if defined(__HIPCC__)
#define GVTM_DESIGN_HD __host__ __device__ inline
#else
#define GVTM_DESIGN_HD inline
#endif

// Output bookkeeping of SampleRateConverter for a track of `steps` internal samples (host and device).
//
// Output sample k is emitted while its integer read position P_k = floor(k * time_inc / 2^16) lies before the end
// pointer; the final end pointer (after flushBuffer()'s 2 * pad zero fills, SampleRateConverter.h:462-471) is
// fills = steps + 2 * pad, so normally N = ceil(fills * 2^16 / time_inc) = k_from.
//
// The flush overrun (SampleRateConverter.h:298-308): when down-sampling the read position advances by more than one
// input per output, so an automatic dataEmpty() (every fill_size = 1024 - 2 * pad fills) can leave the empty pointer
// BEYOND its end pointer.  If fewer fills than that overshoot follow before flushBuffer()'s explicit dataEmpty(), it
// finds endPtr < emptyPtr_, adds BUFFER_SIZE and converts one more lap of the ring: outputs [k_from, k_to) with read
// positions up to fills + 1024, taken from the ring's leftovers.  Returns whether that happens.
GVTM_DESIGN_HD bool src_flush_overrun(unsigned time_inc, int pad, uint64_t steps, uint64_t& k_from, uint64_t& k_to)
{
	const uint64_t fills = steps + 2ull * static_cast<uint64_t>(pad);
	const uint64_t fill_size = static_cast<uint64_t>(kSrcRing - 2 * pad);
	const uint64_t last_auto_end = (fills / fill_size) * fill_size; // end pointer of the last automatic dataEmpty()
	k_from = ((fills << 16) + time_inc - 1) / time_inc;
	k_to = k_from;
	if (last_auto_end == 0) return false;
	const uint64_t k_star = ((last_auto_end << 16) + time_inc - 1) / time_inc; // first output it did not emit
	const uint64_t p_star = (k_star * static_cast<uint64_t>(time_inc)) >> 16; // where it left the empty pointer
	if (p_star <= fills) return false;
	k_to = (((fills + static_cast<uint64_t>(kSrcRing)) << 16) + time_inc - 1) / time_inc;
	return true;
}

// samples in outputBuffer() after finishSynthesis()
GVTM_DESIGN_HD uint64_t src_output_count(unsigned time_inc, int pad, int upsampling, uint64_t steps)
{
	uint64_t k_from = 0, k_to = 0;
	if (upsampling) return (((steps + 2ull * static_cast<uint64_t>(pad)) << 16) + time_inc - 1) / time_inc;
	return src_flush_overrun(time_inc, pad, steps, k_from, k_to) ? k_to : k_from;
}

// the largest src_output_count() over all tracks of at most `steps` internal samples: what a row of a ragged batch
// must be able to hold (a shorter track that runs into the flush overrun can be LONGER than the longest track)
inline uint64_t src_output_capacity(unsigned time_inc, int pad, int upsampling, uint64_t steps)
{
	const uint64_t lap = upsampling ? 0ull : static_cast<uint64_t>(kSrcRing);
	return (((steps + 2ull * static_cast<uint64_t>(pad) + lap) << 16) + time_inc - 1) / time_inc;
}

// --- parameter-track generation (vtm_tracks.hip) ---

// gvtm_track_config with the drift generator's filter designed (DriftGenerator::setUp,
// Butterworth2LowPassFilter<double>::update) — plain data, passed to the kernel by value
struct TrackConstants {
	int control_period;
	int macro_intonation, micro_intonation, intonation_drift, smooth_intonation;
	double initial_pitch, mean_pitch;
	double pitch_deviation, pitch_offset; // deviation * 2, deviation
	double b0, b1, a1, a2;                // Butterworth2LowPassFilter coefficients
};

// "" on success, otherwise what is wrong with the configuration
const char* design_tracks(const gvtm_track_config& cfg, TrackConstants& out);

// frames generateOutput() pushes for one event list (host; the loop over control periods without the arithmetic)
size_t tracks_frame_count(int control_period, const gvtm_event* events, size_t n_events);

} // namespace gvtm
