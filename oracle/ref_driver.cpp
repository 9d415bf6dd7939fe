// TEST INFRASTRUCTURE — not part of the product.
//
// Driver for the *real* GamaTTS vocal-tract models, compiled from the sources
// where they lie under /root/reference (see oracle/Makefile; outputs go to
// oracle/_ref/, which is git-ignored).  This file is our own code: it only
// #includes the reference's public headers at build time.
//
// It replays the per-sample driver loop of
//   GS::VTMControlModel::Controller::synthesize()
//   (gama_tts/src/vtm_control_model/Controller.cpp:277-313)
// over exact float32 parameter frames and writes the float32 output buffer,
// so that goldens never pass through the lossy 6-digit parameter text file
// (SURVEY.md E4).
//
// usage:
//   ref_vtm <config.txt> <model> <output_rate> <control_rate> <params.f32>
//           <n_frames> <out.f32|-> [repeat]
//   <model>: 0,1,2,3,4,5,2000 = VocalTractModel::getInstance factory
//            (gama_tts/src/vtm/VocalTractModel.cpp:35-59);
//            "2:D" = VocalTractModel2<double,D> instantiated directly, D in 1..4
//            "2f:D" = VocalTractModel2<float,D> (no factory number), "4f" = VocalTractModel4<float,1>,
//            "5f" = VocalTractModel5<float,1>
//            "2000:<path>" = plugin factory with dll_path=<path>
//   [repeat] > 1: timing mode, the utterance is synthesised <repeat> times
//            (reset() between runs, as Controller does, Controller.cpp:231).
//   [poll=<nframes>]: the INTERACTIVE caller contract instead (the model is constructed with interactive = true and
//            driven the way gama_tts_editor/src/interactive/InteractiveAudio.cpp:141-185 drives it from its JACK
//            callback): per callback, drain outputBuffer() with Util::getSamples (which clears it when used up), then
//            `while (outputBuffer().size() < needed) { setParameter x 16; execSynthesisStep(); }`; the per-step
//            parameter values are the driver loop's, every callback asks for <nframes> samples; when the steps
//            are used up finishSynthesis() is called and the rest drained.  <out.f32> receives the drained samples
//            (scale 1), which must be the same stream as in the batch protocol.
// prints one line:  N=<samples> steps=<internal steps> fs=<internal rate>
//                   sec=<wall of all repeats> ns_per_step=<...>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ConfigurationData.h"
#include "VTMUtil.h"
#include "VocalTractModel.h"
#include "VocalTractModel2.h"
#include "VocalTractModel4.h"
#include "VocalTractModel5.h"

using GS::ConfigurationData;
using GS::VTM::VocalTractModel;

static std::unique_ptr<VocalTractModel> make_model(ConfigurationData& cfg, const std::string& model, bool interactive = false)
{
	if (model.rfind("2:", 0) == 0) {
		const int d = std::atoi(model.c_str() + 2);
		cfg.put("model", "2");
		cfg.put("log_parameters", "false");
		switch (d) {
		case 1: return std::make_unique<GS::VTM::VocalTractModel2<double, 1>>(cfg, interactive);
		case 2: return std::make_unique<GS::VTM::VocalTractModel2<double, 2>>(cfg, interactive);
		case 3: return std::make_unique<GS::VTM::VocalTractModel2<double, 3>>(cfg, interactive);
		case 4: return std::make_unique<GS::VTM::VocalTractModel2<double, 4>>(cfg, interactive);
		default:
			std::fprintf(stderr, "unsupported SectionDelay %d\n", d);
			std::exit(2);
		}
	}
	if (model.rfind("2f:", 0) == 0) {
		const int d = std::atoi(model.c_str() + 3);
		cfg.put("model", "2");
		cfg.put("log_parameters", "false");
		switch (d) {
		case 1: return std::make_unique<GS::VTM::VocalTractModel2<float, 1>>(cfg, interactive);
		case 2: return std::make_unique<GS::VTM::VocalTractModel2<float, 2>>(cfg, interactive);
		case 3: return std::make_unique<GS::VTM::VocalTractModel2<float, 3>>(cfg, interactive);
		case 4: return std::make_unique<GS::VTM::VocalTractModel2<float, 4>>(cfg, interactive);
		default:
			std::fprintf(stderr, "unsupported SectionDelay %d\n", d);
			std::exit(2);
		}
	}
	if (model == "4f") {
		cfg.put("model", "4");
		cfg.put("log_parameters", "false");
		return std::make_unique<GS::VTM::VocalTractModel4<float, 1>>(cfg, interactive);
	}
	if (model == "5f") {
		cfg.put("model", "5");
		cfg.put("log_parameters", "false");
		return std::make_unique<GS::VTM::VocalTractModel5<float, 1>>(cfg, interactive);
	}
	if (model.rfind("2000:", 0) == 0) {
		cfg.put("model", "2000");
		cfg.put("dll_path", model.c_str() + 5);
		return VocalTractModel::getInstance(cfg, interactive);
	}
	cfg.put("model", model.c_str());
	cfg.put("log_parameters", "false");
	return VocalTractModel::getInstance(cfg, interactive);
}

int main(int argc, char** argv)
{
	if (argc < 8) {
		std::fprintf(stderr, "usage: %s config model output_rate control_rate params.f32 n_frames out.f32|- [repeat]\n", argv[0]);
		return 2;
	}
	try {
		ConfigurationData cfg{argv[1]};
		const std::string model = argv[2];
		cfg.put("output_rate", static_cast<const char*>(argv[3]));
		const double controlRate = std::atof(argv[4]);
		const std::size_t nFrames = std::strtoul(argv[6], nullptr, 10);
		const int repeat = argc > 8 ? std::atoi(argv[8]) : 1;
		const std::size_t pollFrames = (argc > 9 && std::strncmp(argv[9], "poll=", 5) == 0) ? std::strtoul(argv[9] + 5, nullptr, 10) : 0;
		const std::size_t numParam = 16;

		std::vector<std::vector<float>> frames(nFrames, std::vector<float>(numParam));
		{
			FILE* f = std::fopen(argv[5], "rb");
			if (!f) { std::perror(argv[5]); return 2; }
			for (auto& fr : frames) {
				if (std::fread(fr.data(), sizeof(float), numParam, f) != numParam) {
					std::fprintf(stderr, "short read on %s\n", argv[5]);
					return 2;
				}
			}
			std::fclose(f);
		}

		auto vtm = make_model(cfg, model, pollFrames > 0);

		if (pollFrames > 0) {
			// the per-step parameter vectors of the driver loop (Controller.cpp:294-311), handed over one by one
			std::vector<std::vector<float>> stepParams;
			if (!frames.empty()) {
				std::vector<std::vector<float>> list = frames;
				list.push_back(list.back());
				const unsigned int controlSteps = static_cast<unsigned int>(std::rint(vtm->internalSampleRate() / controlRate));
				const float coef = 1.0f / controlSteps;
				std::vector<float> cur(numParam), delta(numParam);
				for (std::size_t i = 1, size = list.size(); i < size; ++i) {
					for (std::size_t j = 0; j < numParam; ++j) {
						cur[j] = list[i - 1][j];
						delta[j] = (list[i][j] - cur[j]) * coef;
					}
					for (std::size_t j = 0; j < controlSteps; ++j) {
						stepParams.push_back(cur);
						for (std::size_t k = 0; k < numParam; ++k) cur[k] += delta[k];
					}
				}
			}
			std::vector<float> drained, callback(pollFrames);
			std::vector<float>& buffer = vtm->outputBuffer();
			std::size_t bufferPos = 0, next = 0, callbacks = 0;
			const auto t0 = std::chrono::steady_clock::now();
			bool finished = false;
			for (;;) {
				// one JACK process() call (InteractiveAudio.cpp:141-204)
				++callbacks;
				std::size_t n = GS::VTM::Util::getSamples(buffer, bufferPos, callback.data(), pollFrames, 1.0f);
				drained.insert(drained.end(), callback.begin(), callback.begin() + n);
				if (n == pollFrames) continue;
				if (finished) break; // nothing more will come
				const std::size_t target = pollFrames - n;
				while (buffer.size() < target && next < stepParams.size()) {
					for (std::size_t i = 0; i < numParam; ++i) vtm->setParameter(static_cast<int>(i), stepParams[next][i]);
					vtm->execSynthesisStep();
					++next;
				}
				if (buffer.size() < target) {
					vtm->finishSynthesis();
					finished = true;
				}
				const std::size_t n2 = GS::VTM::Util::getSamples(buffer, bufferPos, callback.data(), target, 1.0f);
				drained.insert(drained.end(), callback.begin(), callback.begin() + n2);
			}
			const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (std::strcmp(argv[7], "-") != 0) {
				FILE* f = std::fopen(argv[7], "wb");
				if (!f) { std::perror(argv[7]); return 2; }
				std::fwrite(drained.data(), sizeof(float), drained.size(), f);
				std::fclose(f);
			}
			std::printf("N=%zu steps=%zu fs=%.17g sec=%.6f ns_per_step=%.2f callbacks=%zu\n",
					drained.size(), stepParams.size(), vtm->internalSampleRate(), sec, stepParams.empty() ? 0.0 : sec * 1e9 / stepParams.size(), callbacks);
			return 0;
		}

		std::size_t steps = 0;
		const auto t0 = std::chrono::steady_clock::now();
		for (int r = 0; r < repeat; ++r) {
			if (!vtm->outputBuffer().empty()) vtm->reset();
			// --- replay of Controller::synthesize (Controller.cpp:277-313) ---
			std::vector<std::vector<float>> list = frames;
			if (!list.empty()) {
				list.push_back(list.back());
				const unsigned int controlSteps = static_cast<unsigned int>(std::rint(vtm->internalSampleRate() / controlRate));
				const float coef = 1.0f / controlSteps;
				std::vector<float> cur(numParam), delta(numParam);
				for (std::size_t i = 1, size = list.size(); i < size; ++i) {
					for (std::size_t j = 0; j < numParam; ++j) {
						cur[j] = list[i - 1][j];
						delta[j] = (list[i][j] - cur[j]) * coef;
					}
					for (std::size_t j = 0; j < controlSteps; ++j) {
						vtm->setAllParameters(cur);
						vtm->execSynthesisStep();
						for (std::size_t k = 0; k < numParam; ++k) cur[k] += delta[k];
					}
				}
				steps += static_cast<std::size_t>(controlSteps) * nFrames;
			}
			vtm->finishSynthesis();
		}
		const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

		const std::vector<float>& out = vtm->outputBuffer();
		if (std::strcmp(argv[7], "-") != 0) {
			FILE* f = std::fopen(argv[7], "wb");
			if (!f) { std::perror(argv[7]); return 2; }
			std::fwrite(out.data(), sizeof(float), out.size(), f);
			std::fclose(f);
		}
		std::printf("N=%zu steps=%zu fs=%.17g sec=%.6f ns_per_step=%.2f\n",
				out.size(), steps, vtm->internalSampleRate(), sec, steps ? sec * 1e9 / steps : 0.0);
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 1;
	}
	return 0;
}
