#include "batch_controller.hpp"

#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace gvtm {

std::map<std::string, std::string> read_key_value_file(const std::string& path)
{
	std::ifstream in(path, std::ios_base::binary);
	if (!in) throw std::runtime_error("Could not open the file: " + path);
	std::map<std::string, std::string> out;
	std::string line;
	int line_no = 0;
	while (std::getline(in, line)) {
		++line_no;
		if (line.empty() || line[0] == '#') continue;
		const auto eq = line.find('=');
		if (eq == std::string::npos) throw std::runtime_error(path + ": missing separator on line " + std::to_string(line_no));
		auto trim = [](std::string s) {
			const auto b = s.find_first_not_of(" \t\r");
			const auto e = s.find_last_not_of(" \t\r");
			return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
		};
		const std::string key = trim(line.substr(0, eq));
		const std::string value = trim(line.substr(eq + 1));
		if (key.empty() || value.empty()) throw std::runtime_error(path + ": empty key or value on line " + std::to_string(line_no));
		if (!out.emplace(key, value).second) throw std::runtime_error(path + ": duplicate key " + key);
	}
	return out;
}

namespace {

double num(const std::map<std::string, std::string>& k, const char* key)
{
	auto it = k.find(key);
	if (it == k.end()) throw std::runtime_error(std::string("Key '") + key + "' not found.");
	return std::stod(it->second);
}

} // namespace

gvtm_config config_from_keys(const std::map<std::string, std::string>& k, int precision)
{
	gvtm_config c{};
	const int model = k.count("model") ? static_cast<int>(num(k, "model")) : 0;
	switch (model) {
	case 0: case 2: c.section_delay = 1; break;
	case 1: c.section_delay = 1; precision = GVTM_PRECISION_F32; break; // VocalTractModel0<float> (VocalTractModel.cpp:45-46)
	case 3: c.section_delay = 3; break;
	case 4: c.section_delay = 1; c.tube_layout = GVTM_TUBE_30_18; break;
	default:
		throw std::runtime_error("vocal tract model " + std::to_string(model) +
				" is not served by the device path (supported: 0, 1, 2, 3, 4, 5)");
	}
	if (k.count("section_delay")) c.section_delay = static_cast<int>(num(k, "section_delay"));
	c.precision = precision;
	c.output_rate = num(k, "output_rate");
	c.waveform = static_cast<int>(num(k, "waveform"));
	c.noise_modulation = static_cast<int>(num(k, "noise_modulation"));
	c.glottal_pulse_tp = num(k, "glottal_pulse_tp");
	c.glottal_pulse_tn_min = num(k, "glottal_pulse_tn_min");
	c.glottal_pulse_tn_max = num(k, "glottal_pulse_tn_max");
	c.breathiness = num(k, "breathiness");
	c.vocal_tract_length_offset = num(k, "vocal_tract_length_offset");
	c.vocal_tract_length = num(k, "vocal_tract_length");
	c.temperature = num(k, "temperature");
	c.loss_factor = num(k, "loss_factor");
	c.mouth_coefficient = num(k, "mouth_coefficient");
	c.nose_coefficient = num(k, "nose_coefficient");
	c.throat_cutoff = num(k, "throat_cutoff");
	c.throat_volume = num(k, "throat_volume");
	c.mix_offset = num(k, "mix_offset");
	c.global_radius_coef = num(k, "global_radius_coef");
	c.global_nasal_radius_coef = num(k, "global_nasal_radius_coef");
	c.aperture_radius = num(k, "aperture_radius");
	for (int i = 0; i < 5; ++i) c.nasal_radius[i] = num(k, ("nasal_radius_" + std::to_string(i + 1)).c_str());
	for (int i = 0; i < 8; ++i) c.radius_coef[i] = num(k, ("radius_" + std::to_string(i + 1) + "_coef").c_str());
	return c;
}

gvtm5_config config5_from_keys(const std::map<std::string, std::string>& k)
{
	gvtm5_config c{};
	auto flag = [&](const char* key) {
		auto it = k.find(key);
		if (it == k.end()) throw std::runtime_error(std::string("Key '") + key + "' not found.");
		return it->second == "true" || it->second == "1"; // ConfigurationData::convertString<bool>
	};
	c.output_rate = num(k, "output_rate");
	c.waveform = static_cast<int>(num(k, "waveform"));
	c.noise_modulation = static_cast<int>(num(k, "noise_modulation"));
	c.bypass = static_cast<int>(num(k, "bypass"));
	c.constant_radius_mouth_impedance = flag("constant_radius_mouth_impedance") ? 1 : 0;
	c.glottal_pulse_tp = num(k, "glottal_pulse_tp");
	c.glottal_pulse_tn_min = num(k, "glottal_pulse_tn_min");
	c.glottal_pulse_tn_max = num(k, "glottal_pulse_tn_max");
	c.breathiness = num(k, "breathiness");
	c.vocal_tract_length_offset = num(k, "vocal_tract_length_offset");
	c.vocal_tract_length = num(k, "vocal_tract_length");
	c.temperature = num(k, "temperature");
	c.loss_factor = num(k, "loss_factor");
	c.mix_offset = num(k, "mix_offset");
	c.global_radius_coef = num(k, "global_radius_coef");
	c.global_nasal_radius_coef = num(k, "global_nasal_radius_coef");
	for (int i = 0; i < 6; ++i) c.nasal_radius[i] = num(k, ("nasal_radius_" + std::to_string(i + 2)).c_str());
	for (int i = 0; i < 8; ++i) c.radius_coef[i] = num(k, ("radius_" + std::to_string(i + 1) + "_coef").c_str());
	c.glottal_noise_cutoff = num(k, "glottal_noise_cutoff");
	c.frication_noise_cutoff = num(k, "frication_noise_cutoff");
	c.frication_factor = num(k, "frication_factor");
	c.min_glottal_loss = num(k, "min_glottal_loss");
	c.max_glottal_loss = num(k, "max_glottal_loss");
	c.glottal_lowpass_cutoff = num(k, "glottal_lowpass_cutoff");
	if (c.constant_radius_mouth_impedance) c.mouth_impedance_radius = num(k, "mouth_impedance_radius");
	c.precision = GVTM_PRECISION_F64;
	return c;
}

void BatchController::init(const std::map<std::string, std::string>& keys, unsigned control_period_ms, const std::vector<int>& devices, int precision)
{
	if (control_period_ms == 0 || control_period_ms > 4) throw std::runtime_error("Invalid control period."); // VTMControlModelConfiguration.cpp:38
	if (devices.empty()) throw std::runtime_error("no device given");
	const bool model5 = keys.count("model") && static_cast<int>(num(keys, "model")) == 5; // VocalTractModel.cpp:47-48
	gvtm5_config config5{};
	if (model5) {
		config5 = config5_from_keys(keys);
		config_ = gvtm_config{};
		config_.output_rate = config5.output_rate;
	} else {
		config_ = config_from_keys(keys, precision);
	}
	const double control_rate = 1000.0 / control_period_ms;
	for (int device : devices) {
		gvtm_plan* plan = nullptr;
		const int rc = model5 ? gvtm_plan_create_model5(&config5, control_rate, device, &plan) : gvtm_plan_create(&config_, control_rate, device, &plan);
		if (rc != GVTM_OK) {
			const std::string why = gvtm_last_error();
			for (gvtm_plan* p : plans_) gvtm_plan_destroy(p);
			plans_.clear();
			throw std::runtime_error(why);
		}
		plans_.push_back(plan);
	}
}

void BatchController::loadVoice(const std::string& voice_dir, const std::vector<int>& devices, int precision)
{
	const std::string dir = (!voice_dir.empty() && voice_dir.back() == '/') ? voice_dir : voice_dir + '/';
	const auto index = read_key_value_file(dir + "_index.txt");
	auto entry = [&](const char* key) {
		auto it = index.find(key);
		if (it == index.end()) throw std::runtime_error(std::string("Key '") + key + "' not found in " + dir + "_index.txt");
		return dir + it->second;
	};
	auto keys = read_key_value_file(entry("vtm_file"));
	const auto control = read_key_value_file(entry("vtm_control_model_file"));
	auto vn = control.find("variant_name");
	if (vn == control.end()) throw std::runtime_error("Key 'variant_name' not found.");
	for (const auto& kv : read_key_value_file(entry("variant_dir") + vn->second + ".txt")) keys[kv.first] = kv.second; // insert(): overwrite
	init(keys, static_cast<unsigned>(num(control, "control_period")), devices, precision);
}

BatchController::BatchController(const std::string& voice_dir, int device, int precision)
{
	loadVoice(voice_dir, std::vector<int>{device}, precision);
}

BatchController::BatchController(const std::string& voice_dir, const std::vector<int>& devices, int precision)
{
	loadVoice(voice_dir, devices, precision);
}

BatchController::BatchController(const std::map<std::string, std::string>& merged_keys, unsigned control_period_ms, int device, int precision)
{
	init(merged_keys, control_period_ms, std::vector<int>{device}, precision);
}

BatchController::~BatchController()
{
	params_buf_.release();
	audio_buf_.release();
	pcm_buf_.release();
	for (gvtm_plan* p : plans_) gvtm_plan_destroy(p);
}

double BatchController::internalSampleRate() const
{
	gvtm_info info{};
	gvtm_plan_info(plans_.front(), &info);
	return info.internal_rate_hz;
}

std::size_t BatchController::addUtteranceFromStream(std::istream& in)
{
	std::vector<float> frames;
	std::string line;
	unsigned line_no = 1;
	while (std::getline(in, line)) {
		std::istringstream ls(line);
		float v[GVTM_N_PARAM];
		for (float& x : v) ls >> x;
		if (!ls) {
			throw std::runtime_error("Could not read vocal tract parameters from stream (line number " + std::to_string(line_no) + ").");
		}
		frames.insert(frames.end(), v, v + GVTM_N_PARAM);
		++line_no;
	}
	return addUtterance(std::move(frames));
}

std::size_t BatchController::addUtterance(std::vector<float> frames)
{
	if (frames.size() % GVTM_N_PARAM != 0) throw std::runtime_error("parameter frames must hold 16 values each");
	utterances_.push_back(std::move(frames));
	return utterances_.size() - 1;
}

void BatchController::HostBuffer::release()
{
	if (ptr) {
		if (pinned) gvtm_host_free(ptr);
		else std::free(ptr);
	}
	ptr = nullptr;
	bytes = 0;
	pinned = false;
}

void BatchController::HostBuffer::resize(std::size_t need)
{
	if (need <= bytes) return;
	release();
	void* p = nullptr;
	if (gvtm_host_alloc(need, &p) == GVTM_OK && p) {
		pinned = true;
	} else {
		p = std::malloc(need);
		if (!p) throw std::bad_alloc();
	}
	ptr = p;
	bytes = need;
}

void BatchController::synthesize(Output output)
{
	const std::size_t batch = utterances_.size();
	shards_.assign(plans_.size(), {0, 0});
	output_ = output;
	if (batch == 0) return;
	std::size_t max_frames = 0;
	std::vector<int32_t> frames(batch);
	for (std::size_t b = 0; b < batch; ++b) {
		frames[b] = static_cast<int32_t>(utterances_[b].size() / GVTM_N_PARAM);
		max_frames = std::max<std::size_t>(max_frames, static_cast<std::size_t>(frames[b]));
	}
	const std::size_t row_in = max_frames * GVTM_N_PARAM;
	params_buf_.resize(sizeof(float) * std::max<std::size_t>(batch * row_in, 1));
	float* const params = static_cast<float*>(params_buf_.ptr);
	for (std::size_t b = 0; b < batch; ++b) {
		float* row = params + b * row_in;
		std::copy(utterances_[b].begin(), utterances_[b].end(), row);
		std::fill(row + utterances_[b].size(), row + row_in, 0.0f);
	}
	// row stride of a ragged batch: on a down-sampling plan a SHORTER utterance that runs into the converter's flush overrun
	// yields more samples than the longest one (include/gama_vtm.h, gvtm_output_capacity)
	stride_ = gvtm_output_capacity(plans_.front(), max_frames);
	if (stride_ == static_cast<std::size_t>(-1)) throw std::runtime_error(gvtm_last_error());
	const bool pcm16 = output == Output::Pcm16;
	if (pcm16) pcm_buf_.resize(sizeof(int16_t) * std::max<std::size_t>(batch * stride_, 1));
	else audio_buf_.resize(sizeof(float) * std::max<std::size_t>(batch * stride_, 1));
	float* const audio = static_cast<float*>(audio_buf_.ptr);
	int16_t* const pcm = static_cast<int16_t*>(pcm_buf_.ptr);
	counts_.assign(batch, 0);
	maxabs_.assign(batch, 0.0f);
	// contiguous shards whose sizes differ by at most one; every shard on its own host thread and device
	const std::size_t n_dev = plans_.size();
	const std::size_t base = batch / n_dev, extra = batch % n_dev;
	std::vector<std::string> errors(n_dev);
	std::vector<std::thread> workers;
	std::size_t lo = 0;
	for (std::size_t d = 0; d < n_dev; ++d) {
		const std::size_t hi = lo + base + (d < extra ? 1 : 0);
		shards_[d] = {lo, hi};
		if (hi > lo) {
			auto run = [this, d, lo, hi, max_frames, row_in, params, audio, pcm, pcm16, &frames, &errors]() {
				const int rc = pcm16
					? gvtm_synthesize_batch_host_pcm16(plans_[d], params + lo * row_in, frames.data() + lo, hi - lo, max_frames,
							pcm + lo * stride_, stride_, counts_.data() + lo, maxabs_.data() + lo, nullptr)
					: gvtm_synthesize_batch_host(plans_[d], params + lo * row_in, frames.data() + lo, hi - lo, max_frames,
							audio + lo * stride_, stride_, counts_.data() + lo, maxabs_.data() + lo);
				if (rc != GVTM_OK) errors[d] = gvtm_last_error(); // thread-local message
			};
			if (n_dev == 1) run();
			else workers.emplace_back(run);
		}
		lo = hi;
	}
	for (auto& w : workers) w.join();
	for (const auto& e : errors) {
		if (!e.empty()) throw std::runtime_error("synthesis failed: " + e);
	}
}

const float* BatchController::samples(std::size_t i) const
{
	if (output_ != Output::Float32 || !audio_buf_.ptr) throw std::logic_error("samples(): the last synthesize() did not bring float samples back");
	return static_cast<const float*>(audio_buf_.ptr) + i * stride_;
}

const int16_t* BatchController::pcm(std::size_t i) const
{
	if (output_ != Output::Pcm16 || !pcm_buf_.ptr) throw std::logic_error("pcm(): the last synthesize() did not bring 16-bit samples back");
	return static_cast<const int16_t*>(pcm_buf_.ptr) + i * stride_;
}

std::size_t BatchController::sampleCount(std::size_t i) const
{
	// (the device clips its writes at the row stride; a count beyond it would be a sizing error above, never a read past the row)
	return std::min(static_cast<std::size_t>(counts_.at(i)), stride_);
}

float BatchController::outputScale(std::size_t i) const
{
	const float peak = maxabs_.at(i);
	return peak < 1.0e-30f ? 0.0f : 0.95f / peak;
}

std::vector<float> BatchController::scaledBuffer(std::size_t i) const
{
	const float scale = outputScale(i);
	std::vector<float> out(sampleCount(i));
	const float* x = samples(i);
	for (std::size_t n = 0; n < out.size(); ++n) out[n] = x[n] * scale;
	return out;
}

void BatchController::writeWav(std::size_t i, const std::string& path) const
{
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw std::runtime_error("Could not open the file " + path + " for writing.");
	auto u32 = [&](int v) { const unsigned char a[4] = {static_cast<unsigned char>(v & 0xff), static_cast<unsigned char>((v >> 8) & 0xff),
			static_cast<unsigned char>((v >> 16) & 0xff), static_cast<unsigned char>((v >> 24) & 0xff)}; std::fwrite(a, 1, 4, f); };
	auto u16 = [&](int v) { const unsigned char a[2] = {static_cast<unsigned char>(v & 0xff), static_cast<unsigned char>((v >> 8) & 0xff)}; std::fwrite(a, 1, 2, f); };
	// header fields as WAVEFileWriter::writeWaveFileHeader (WAVEFileWriter.cpp:62-118), mono 16-bit PCM
	const int n = static_cast<int>(sampleCount(i));
	const int data_bytes = n * 2;
	const float rate = static_cast<float>(config_.output_rate);
	std::fputs("RIFF", f); u32(4 + 24 + (8 + data_bytes)); std::fputs("WAVE", f);
	std::fputs("fmt ", f); u32(16); u16(1); u16(1);
	u32(static_cast<int>(std::round(rate))); u32(static_cast<int>(std::ceil(rate * 2))); u16(2); u16(16);
	std::fputs("data", f); u32(data_bytes);
	if (output_ == Output::Pcm16) {
		// scaled and rounded on the device by the same rule (vtm_normalize_kernel)
		const int16_t* x = pcm(i);
		for (int s = 0; s < n; ++s) u16(static_cast<int>(x[s]));
	} else {
		const float scale = outputScale(i);
		const float* x = samples(i);
		for (int s = 0; s < n; ++s) u16(static_cast<int>(std::round((x[s] * scale) * 32767.0f))); // writeSample, WAVEFileWriter.cpp:122-125
	}
	std::fclose(f);
}

} // namespace gvtm
