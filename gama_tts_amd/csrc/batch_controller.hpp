// Batched counterpart of the VTM-driving half of GS::VTMControlModel::Controller
// (gama_tts/src/vtm_control_model/Controller.{h,cpp}): it takes the same inputs — a voice
// directory (or a merged key=value configuration) and parameter streams with one 16-float
// frame per line — and produces what Controller::synthesizeToFile / synthesizeToBuffer produce,
// for many utterances at once, through the C ABI of libgama_vtm.so.
#pragma once

#include <cstdint>
#include <iosfwd>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gama_vtm.h"

namespace gvtm {

// key = value file, '#' comments at line start (ConfigurationData.cpp:67-118 format).
std::map<std::string, std::string> read_key_value_file(const std::string& path);

// vtm.txt keys (merged with the variant) -> gvtm_config.  `model` selects the semantics:
// 0, 2 -> SectionDelay 1; 3 -> SectionDelay 3; 4 -> the 30+18-section tube (VocalTractModel.cpp:40-49);
// 1 -> model 0 in float; 5 is handled by config5_from_keys; others are refused.
gvtm_config config_from_keys(const std::map<std::string, std::string>& keys, int precision);
// the keys of a model-5 voice (VocalTractModel5::loadConfiguration, vtm/VocalTractModel5.h:375-421) -> gvtm5_config
gvtm5_config config5_from_keys(const std::map<std::string, std::string>& keys);

class BatchController {
public:
	// Loads <voice_dir>/_index.txt, the vtm file, the control-model file (control_period,
	// variant_name) and the variant, merged the way Controller's constructor does
	// (Controller.cpp:40-55, VTMControlModelConfiguration.cpp:32-60).
	BatchController(const std::string& voice_dir, int device, int precision = GVTM_PRECISION_F64);
	// From an already merged configuration and an explicit control period in ms (1..4).
	BatchController(const std::map<std::string, std::string>& merged_keys, unsigned control_period_ms, int device,
			int precision = GVTM_PRECISION_F64);
	// Several devices of one node (BASELINE configs[4]): the queued utterances are cut into contiguous shards,
	// one per listed device, each synthesized by its own host thread through its own plan and stream; no
	// exchange between devices (utterances are independent).  A device may be listed twice.
	BatchController(const std::string& voice_dir, const std::vector<int>& devices, int precision = GVTM_PRECISION_F64);
	~BatchController();
	BatchController(const BatchController&) = delete;
	BatchController& operator=(const BatchController&) = delete;

	// One frame per line, 16 whitespace-separated floats (Controller::getParametersFromStream,
	// Controller.cpp:170-192).  Returns the utterance index.
	std::size_t addUtteranceFromStream(std::istream& in);
	std::size_t addUtterance(std::vector<float> frames /* [n][16] */);
	std::size_t size() const { return utterances_.size(); }

	// What synthesize() brings back from the device:
	//   Float32  the unscaled outputBuffer() samples (samples(), scaledBuffer(), writeWav() all work)
	//   Pcm16    the int16 samples Controller::writeOutputToFile would put into the WAV file, scaled and rounded on the
	//            device (half the bytes over PCIe; pcm() and writeWav() work) -- what the batched CLI uses
	enum class Output { Float32, Pcm16 };
	// Controller::synthesize + finishSynthesis for every queued utterance (one launch per device, or a pipeline of slices
	// for big batches: include/gama_vtm.h, gvtm_synthesize_batch_host*).  Host buffers are page-locked.
	void synthesize(Output output = Output::Float32);
	std::size_t deviceCount() const { return plans_.size(); }
	// [first, last) utterance indices handed to device slot `d` by the last synthesize()
	std::pair<std::size_t, std::size_t> shard(std::size_t d) const { return shards_.at(d); }

	double outputSampleRate() const { return config_.output_rate; }
	double internalSampleRate() const;
	// Unscaled samples of utterance i (VocalTractModel::outputBuffer()); Output::Float32 only.
	const float* samples(std::size_t i) const;
	// The utterance's 16-bit samples as the WAV file holds them; Output::Pcm16 only.
	const int16_t* pcm(std::size_t i) const;
	std::size_t sampleCount(std::size_t i) const;
	// Util::calculateOutputScale: 0.95 / max|x|, 0 below 1e-30 (VTMUtil.cpp:48-67).
	float outputScale(std::size_t i) const;
	// Controller::writeOutputToBuffer (Controller.cpp:330-340).
	std::vector<float> scaledBuffer(std::size_t i) const;
	// Controller::writeOutputToFile (Controller.cpp:315-328): 16-bit mono RIFF/WAVE.
	void writeWav(std::size_t i, const std::string& path) const;
private:
	void init(const std::map<std::string, std::string>& keys, unsigned control_period_ms, const std::vector<int>& devices, int precision);
	void loadVoice(const std::string& voice_dir, const std::vector<int>& devices, int precision);
	gvtm_config config_{};
	std::vector<gvtm_plan*> plans_; // one per device slot
	std::vector<std::pair<std::size_t, std::size_t>> shards_;
	std::vector<std::vector<float>> utterances_;
	// page-locked host buffers (gvtm_host_alloc; plain memory when that fails), freed by the destructor
	struct HostBuffer {
		void* ptr = nullptr;
		std::size_t bytes = 0;
		bool pinned = false;
		void resize(std::size_t need);
		void release();
	};
	HostBuffer params_buf_, audio_buf_, pcm_buf_;
	Output output_ = Output::Float32;
	std::vector<int64_t> counts_;
	std::vector<float> maxabs_;
	std::size_t stride_ = 0;
};

} // namespace gvtm
