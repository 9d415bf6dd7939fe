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
#include "VocalTractModel.h"
#include "VocalTractModel2.h"
#include "VocalTractModel4.h"
#include "VocalTractModel5.h"

using GS::ConfigurationData;
using GS::VTM::VocalTractModel;

static std::unique_ptr<VocalTractModel> make_model(ConfigurationData& cfg, const std::string& model)
{
	if (model.rfind("2:", 0) == 0) {
		const int d = std::atoi(model.c_str() + 2);
		cfg.put("model", "2");
		cfg.put("log_parameters", "false");
		switch (d) {
		case 1: return std::make_unique<GS::VTM::VocalTractModel2<double, 1>>(cfg, false);
		case 2: return std::make_unique<GS::VTM::VocalTractModel2<double, 2>>(cfg, false);
		case 3: return std::make_unique<GS::VTM::VocalTractModel2<double, 3>>(cfg, false);
		case 4: return std::make_unique<GS::VTM::VocalTractModel2<double, 4>>(cfg, false);
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
		case 1: return std::make_unique<GS::VTM::VocalTractModel2<float, 1>>(cfg, false);
		case 2: return std::make_unique<GS::VTM::VocalTractModel2<float, 2>>(cfg, false);
		case 3: return std::make_unique<GS::VTM::VocalTractModel2<float, 3>>(cfg, false);
		case 4: return std::make_unique<GS::VTM::VocalTractModel2<float, 4>>(cfg, false);
		default:
			std::fprintf(stderr, "unsupported SectionDelay %d\n", d);
			std::exit(2);
		}
	}
	if (model == "4f") {
		cfg.put("model", "4");
		cfg.put("log_parameters", "false");
		return std::make_unique<GS::VTM::VocalTractModel4<float, 1>>(cfg, false);
	}
	if (model == "5f") {
		cfg.put("model", "5");
		cfg.put("log_parameters", "false");
		return std::make_unique<GS::VTM::VocalTractModel5<float, 1>>(cfg, false);
	}
	if (model.rfind("2000:", 0) == 0) {
		cfg.put("model", "2000");
		cfg.put("dll_path", model.c_str() + 5);
		return VocalTractModel::getInstance(cfg);
	}
	cfg.put("model", model.c_str());
	cfg.put("log_parameters", "false");
	return VocalTractModel::getInstance(cfg);
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

		auto vtm = make_model(cfg, model);

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
