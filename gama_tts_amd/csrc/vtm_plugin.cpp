// libgama_vtm_plugin.so — GamaTTS VocalTractModel plugin over the C ABI of libgama_vtm.so.
// See include/gama_vtm_plugin.h for the contract.  Host-side C++ only; all synthesis happens
// in gvtm_synthesize_batch_host() (batch-friendly protocol: the steps are recorded and synthesized at
// finishSynthesis()) or in a gvtm_stream (interactive protocol: outputBuffer() is polled after every
// execSynthesisStep(), gama_tts_editor/src/interactive/InteractiveAudio.cpp:141-185, so the steps go to the
// device in blocks and the samples appear in bursts, as the reference's own converter delivers them every
// 998 steps, vtm/SampleRateConverter.h:277-281).
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gama_vtm.h"
#include "../../include/gama_vtm_plugin.h"

namespace {

// Layout mirror of GS::ConfigurationData (gama_tts/src/ConfigurationData.h:58-65).
struct ConfigurationDataMirror {
	std::string filePath;
	std::string dirPath;
	std::unordered_map<std::string, std::string> valueMap;
};

// Same virtual-function order as GS::VTM::VocalTractModel (vtm/VocalTractModel.h:46-59).
class VocalTractModelAbi {
public:
	virtual ~VocalTractModelAbi() noexcept = default;
	virtual void reset() noexcept = 0;
	virtual double internalSampleRate() const noexcept = 0;
	virtual double outputSampleRate() const noexcept = 0;
	virtual void setParameter(int parameter, float value) noexcept = 0;
	virtual void setAllParameters(const std::vector<float>& parameters) noexcept = 0;
	virtual void execSynthesisStep() noexcept = 0;
	virtual void finishSynthesis() noexcept = 0;
	virtual std::vector<float>& outputBuffer() noexcept = 0;
};

class KeyReader {
public:
	explicit KeyReader(const ConfigurationDataMirror& d) : d_(d) {}
	// Same conversions as ConfigurationData::convertString<double/int> (ConfigurationData.cpp:122-166).
	double number(const char* key) const
	{
		auto it = d_.valueMap.find(key);
		if (it == d_.valueMap.end()) throw std::runtime_error(std::string("Key '") + key + "' not found in " + d_.filePath);
		return std::stod(it->second);
	}
	int integer(const char* key) const
	{
		auto it = d_.valueMap.find(key);
		if (it == d_.valueMap.end()) throw std::runtime_error(std::string("Key '") + key + "' not found in " + d_.filePath);
		return std::stoi(it->second);
	}
	std::string text(const char* key, const char* fallback) const
	{
		auto it = d_.valueMap.find(key);
		return it == d_.valueMap.end() ? std::string(fallback) : it->second;
	}
	bool has(const char* key) const { return d_.valueMap.count(key) != 0; }
private:
	const ConfigurationDataMirror& d_;
};

class DeviceVocalTractModel final : public VocalTractModelAbi {
public:
	DeviceVocalTractModel(const ConfigurationDataMirror& data, bool interactive)
		: interactive_(interactive)
	{
		const KeyReader k(data);
		const int device = k.has("gpu_device") ? k.integer("gpu_device") : 0;
		if (k.has("gpu_model") && k.integer("gpu_model") == 5) {
			// the keys of VocalTractModel5::loadConfiguration (vtm/VocalTractModel5.h:375-421)
			gvtm5_config c5{};
			c5.output_rate = k.number("output_rate");
			c5.waveform = k.integer("waveform");
			c5.noise_modulation = k.integer("noise_modulation");
			c5.bypass = k.integer("bypass");
			const std::string cm = k.text("constant_radius_mouth_impedance", "false");
			c5.constant_radius_mouth_impedance = (cm == "true" || cm == "1") ? 1 : 0;
			c5.glottal_pulse_tp = k.number("glottal_pulse_tp");
			c5.glottal_pulse_tn_min = k.number("glottal_pulse_tn_min");
			c5.glottal_pulse_tn_max = k.number("glottal_pulse_tn_max");
			c5.breathiness = k.number("breathiness");
			c5.vocal_tract_length_offset = k.number("vocal_tract_length_offset");
			c5.vocal_tract_length = k.number("vocal_tract_length");
			c5.temperature = k.number("temperature");
			c5.loss_factor = k.number("loss_factor");
			c5.mix_offset = k.number("mix_offset");
			c5.global_radius_coef = k.number("global_radius_coef");
			c5.global_nasal_radius_coef = k.number("global_nasal_radius_coef");
			static const char* const nasal5[6] = {"nasal_radius_2", "nasal_radius_3", "nasal_radius_4", "nasal_radius_5", "nasal_radius_6", "nasal_radius_7"};
			static const char* const coef5[8] = {"radius_1_coef", "radius_2_coef", "radius_3_coef", "radius_4_coef",
					"radius_5_coef", "radius_6_coef", "radius_7_coef", "radius_8_coef"};
			for (int i = 0; i < 6; ++i) c5.nasal_radius[i] = k.number(nasal5[i]);
			for (int i = 0; i < 8; ++i) c5.radius_coef[i] = k.number(coef5[i]);
			c5.glottal_noise_cutoff = k.number("glottal_noise_cutoff");
			c5.frication_noise_cutoff = k.number("frication_noise_cutoff");
			c5.frication_factor = k.number("frication_factor");
			c5.min_glottal_loss = k.number("min_glottal_loss");
			c5.max_glottal_loss = k.number("max_glottal_loss");
			c5.glottal_lowpass_cutoff = k.number("glottal_lowpass_cutoff");
			if (c5.constant_radius_mouth_impedance) c5.mouth_impedance_radius = k.number("mouth_impedance_radius");
			c5.precision = GVTM_PRECISION_F64;
			gvtm_plan* probe = nullptr;
			if (gvtm_plan_create_model5(&c5, 1000.0, GVTM_DEVICE_NONE, &probe) != GVTM_OK) throw std::runtime_error(gvtm_last_error());
			gvtm_info info{};
			gvtm_plan_info(probe, &info);
			gvtm_plan_destroy(probe);
			internal_rate_ = info.internal_rate_hz; // not an integer (vtm/VocalTractModel5.h:465)
			output_rate_ = c5.output_rate;
			if (gvtm_plan_create_model5(&c5, info.internal_rate_hz, device, &plan_) != GVTM_OK) throw std::runtime_error(gvtm_last_error());
			gvtm_plan_info(plan_, &info);
			if (info.control_steps != 1) throw std::runtime_error("internal error: plugin plan must run one step per frame");
			current_.assign(GVTM_N_PARAM, 0.0f);
			output_.reserve(1024);
			if (interactive_) startInteractive(k);
			return;
		}
		gvtm_config c{};
		c.output_rate = k.number("output_rate");
		c.waveform = k.integer("waveform");
		c.noise_modulation = k.integer("noise_modulation");
		c.glottal_pulse_tp = k.number("glottal_pulse_tp");
		c.glottal_pulse_tn_min = k.number("glottal_pulse_tn_min");
		c.glottal_pulse_tn_max = k.number("glottal_pulse_tn_max");
		c.breathiness = k.number("breathiness");
		c.vocal_tract_length_offset = k.number("vocal_tract_length_offset");
		c.vocal_tract_length = k.number("vocal_tract_length");
		c.temperature = k.number("temperature");
		c.loss_factor = k.number("loss_factor");
		c.mouth_coefficient = k.number("mouth_coefficient");
		c.nose_coefficient = k.number("nose_coefficient");
		c.throat_cutoff = k.number("throat_cutoff");
		c.throat_volume = k.number("throat_volume");
		c.mix_offset = k.number("mix_offset");
		c.global_radius_coef = k.number("global_radius_coef");
		c.global_nasal_radius_coef = k.number("global_nasal_radius_coef");
		c.aperture_radius = k.number("aperture_radius");
		static const char* const nasal_keys[5] = {"nasal_radius_1", "nasal_radius_2", "nasal_radius_3", "nasal_radius_4", "nasal_radius_5"};
		static const char* const coef_keys[8] = {"radius_1_coef", "radius_2_coef", "radius_3_coef", "radius_4_coef",
				"radius_5_coef", "radius_6_coef", "radius_7_coef", "radius_8_coef"};
		for (int i = 0; i < 5; ++i) c.nasal_radius[i] = k.number(nasal_keys[i]);
		for (int i = 0; i < 8; ++i) c.radius_coef[i] = k.number(coef_keys[i]);
		c.section_delay = k.has("section_delay") ? k.integer("section_delay") : 1;
		c.tube_layout = k.has("tube_layout") ? k.integer("tube_layout") : GVTM_TUBE_10_6;
		const std::string prec = k.text("gpu_precision", "f64");
		c.precision = prec == "mixed" ? GVTM_PRECISION_MIXED : (prec == "f32" ? GVTM_PRECISION_F32 : GVTM_PRECISION_F64);

		// The host interpolates the control frames itself (Controller.cpp:294-311) and hands over
		// one parameter vector per internal step, so the plan runs with one step per "frame":
		// control rate == internal sample rate.
		gvtm_plan* probe = nullptr;
		if (gvtm_plan_create(&c, 1000.0, GVTM_DEVICE_NONE, &probe) != GVTM_OK) throw std::runtime_error(gvtm_last_error());
		gvtm_info info{};
		gvtm_plan_info(probe, &info);
		gvtm_plan_destroy(probe);
		internal_rate_ = info.internal_sample_rate;
		output_rate_ = c.output_rate;
		if (gvtm_plan_create(&c, static_cast<double>(info.internal_sample_rate), device, &plan_) != GVTM_OK) {
			throw std::runtime_error(gvtm_last_error());
		}
		gvtm_plan_info(plan_, &info);
		if (info.control_steps != 1) throw std::runtime_error("internal error: plugin plan must run one step per frame");
		current_.assign(GVTM_N_PARAM, 0.0f);
		output_.reserve(1024);
		if (interactive_) startInteractive(k);
	}

	// interactive protocol (is_interactive = true, VocalTractModelPlugin.cpp:87; InteractiveAudio.cpp:141-185): the recorded
	// steps go to a stream in blocks
	void startInteractive(const KeyReader& k)
	{
		// steps per launch: the reference's converter hands its samples over every 998 fills; a multiple of 12 near it
		// (12 steps: what the serial wavefronts of every model's kernel are exact at)
		int block = k.has("gpu_interactive_block") ? k.integer("gpu_interactive_block") : 996;
		block_steps_ = static_cast<std::size_t>(block < 12 ? 12 : ((block + 11) / 12) * 12);
		if (gvtm_stream_create(plan_, 1, &stream_) != GVTM_OK) {
			const std::string why = gvtm_last_error();
			gvtm_plan_destroy(plan_);
			plan_ = nullptr;
			throw std::runtime_error(why);
		}
		burst_.resize(gvtm_stream_capacity(stream_, block_steps_));
	}
	~DeviceVocalTractModel() noexcept override
	{
		gvtm_stream_destroy(stream_);
		gvtm_plan_destroy(plan_);
	}

	void reset() noexcept override
	{
		steps_.clear();
		output_.clear();
		failed_ = false;
		if (stream_ && gvtm_stream_reset(stream_) != GVTM_OK) {
			std::fprintf(stderr, "[gama_vtm_plugin] reset failed: %s\n", gvtm_last_error());
			failed_ = true;
		}
	}
	double internalSampleRate() const noexcept override { return internal_rate_; }
	double outputSampleRate() const noexcept override { return output_rate_; }
	void setParameter(int parameter, float value) noexcept override
	{
		if (parameter < 0 || parameter >= GVTM_N_PARAM) return; // fail silently, VocalTractModel0.h:690-692
		current_[static_cast<std::size_t>(parameter)] = value;
	}
	void setAllParameters(const std::vector<float>& parameters) noexcept override
	{
		if (parameters.size() != GVTM_N_PARAM) return; // fail silently, VocalTractModel0.h:700-703
		current_ = parameters;
	}
	void execSynthesisStep() noexcept override
	{
		try {
			steps_.insert(steps_.end(), current_.begin(), current_.end());
			if (interactive_ && steps_.size() >= block_steps_ * GVTM_N_PARAM) pushBlock();
		} catch (...) {
			failed_ = true;
		}
	}
	void finishSynthesis() noexcept override
	{
		// Whatever happens, the next utterance starts from a clean recording: the reference's Controller only calls
		// reset() when outputBuffer() is not empty (Controller.cpp:231), and a failed utterance leaves it empty.
		try {
			if (interactive_) {
				finishStream();
			} else {
				finishRecorded();
			}
		} catch (...) {
			output_.clear();
		}
		steps_.clear();
		failed_ = false;
	}
	std::vector<float>& outputBuffer() noexcept override { return output_; }
private:
	void finishRecorded()
	{
		const std::size_t n_steps = steps_.size() / GVTM_N_PARAM;
		const std::size_t n_out = gvtm_output_count(plan_, n_steps);
		if (failed_ || n_out == static_cast<std::size_t>(-1)) {
			std::fprintf(stderr, "[gama_vtm_plugin] cannot synthesize: %s\n", failed_ ? "out of memory while recording" : gvtm_last_error());
			output_.clear();
			return;
		}
		output_.assign(n_out, 0.0f);
		int64_t written = 0;
		const int rc = gvtm_synthesize_batch_host(plan_, steps_.data(), nullptr, 1, n_steps, output_.data(), n_out, &written, nullptr);
		if (rc != GVTM_OK || written < 0) {
			std::fprintf(stderr, "[gama_vtm_plugin] synthesis failed: %s\n", gvtm_last_error());
			output_.clear();
			return;
		}
		output_.resize(static_cast<std::size_t>(written));
	}
	// interactive protocol: the recorded steps go to the stream, the samples it returns are appended to the buffer the
	// host is polling (the host may have emptied it in between: Util::getSamples, vtm/VTMUtil.cpp:41-44)
	void pushBlock()
	{
		if (failed_) { steps_.clear(); return; }
		const std::size_t n_steps = steps_.size() / GVTM_N_PARAM;
		const std::size_t cap = gvtm_stream_capacity(stream_, n_steps);
		if (burst_.size() < cap) burst_.resize(cap);
		int64_t n = 0;
		if (gvtm_stream_push(stream_, steps_.data(), nullptr, n_steps, burst_.data(), burst_.size(), &n) != GVTM_OK) {
			std::fprintf(stderr, "[gama_vtm_plugin] synthesis failed: %s\n", gvtm_last_error());
			failed_ = true;
		} else {
			output_.insert(output_.end(), burst_.begin(), burst_.begin() + n);
		}
		steps_.clear();
	}
	void finishStream()
	{
		if (!steps_.empty()) pushBlock();
		if (!failed_) {
			const std::size_t cap = gvtm_stream_capacity(stream_, 0);
			if (burst_.size() < cap) burst_.resize(cap);
			int64_t n = 0;
			if (gvtm_stream_finish(stream_, burst_.data(), burst_.size(), &n, nullptr) != GVTM_OK) {
				std::fprintf(stderr, "[gama_vtm_plugin] synthesis failed: %s\n", gvtm_last_error());
				failed_ = true;
			} else {
				output_.insert(output_.end(), burst_.begin(), burst_.begin() + n);
			}
		}
		if (failed_) output_.clear();
		// Every finishSynthesis(), failed or not, leaves a clean start: the host only calls reset() when outputBuffer() is
		// non-empty (Controller.cpp:231), so a failed utterance must not leave its held frames and device state behind.
		if (gvtm_stream_reset(stream_) != GVTM_OK) failed_ = true;
	}

	const bool interactive_;
	gvtm_stream* stream_ = nullptr;
	std::size_t block_steps_ = 996;
	std::vector<float> burst_;
	gvtm_plan* plan_ = nullptr;
	double internal_rate_ = 0.0;
	double output_rate_ = 0.0;
	bool failed_ = false;
	std::vector<float> current_;
	std::vector<float> steps_;  // [n_steps][16], as handed over by the host
	std::vector<float> output_;
};

} // namespace

extern "C" {

void* GAMA_TTS_construct_vocal_tract_model(const void* config_data, int is_interactive)
{
	if (!config_data) return nullptr;
	try {
		return static_cast<VocalTractModelAbi*>(new DeviceVocalTractModel(*static_cast<const ConfigurationDataMirror*>(config_data), is_interactive != 0));
	} catch (const std::exception& e) {
		std::fprintf(stderr, "[gama_vtm_plugin] construct failed: %s\n", e.what());
	} catch (...) {
		std::fprintf(stderr, "[gama_vtm_plugin] construct failed\n");
	}
	return nullptr;
}

void GAMA_TTS_destruct_vocal_tract_model(void* vtm)
{
	delete static_cast<VocalTractModelAbi*>(vtm);
}

} // extern "C"
