// TEST INFRASTRUCTURE — not part of the product.
//
// Captures the exact float32 VTM parameter frames the real GamaTTS text->posture
// pipeline produces for a text (BASELINE.json configs[0], "Hello world"), so that
// a real track can be committed as a fixture without passing through the lossy
// 6-digit `-p` dump.  Our own code; it mirrors what tts() in
// gama_tts/src/main.cpp:106-194 does and reads Controller::vtmParameterList()
// (vtm_control_model/Controller.h:54).  Compiled against the reference where it
// lies (oracle/Makefile target ref_full); never shipped.
//
// usage: ref_tts_capture <voice_data_dir> "<text>" <out_params.f32>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "Controller.h"
#include "Index.h"
#include "Model.h"
#include "TextParser.h"

int main(int argc, char** argv)
{
	if (argc != 4) {
		std::fprintf(stderr, "usage: %s voice_dir text out_params.f32\n", argv[0]);
		return 2;
	}
	try {
		const GS::Index index{argv[1]};
		auto model = std::make_unique<GS::VTMControlModel::Model>();
		model->load(index);
		auto controller = std::make_unique<GS::VTMControlModel::Controller>(index, *model);
		auto parser = GS::TextParser::TextParser::getInstance(index, controller->vtmControlModelConfiguration().phoStrFormat);
		const std::string pho = parser->parse(argv[2]);
		std::vector<float> audio;
		controller->synthesizePhoneticStringToBuffer(pho, nullptr, audio);
		const auto& list = controller->vtmParameterList();
		// Controller::synthesize appended a copy of the last frame (Controller.cpp:283).
		const std::size_t frames = list.empty() ? 0 : list.size() - 1;
		FILE* f = std::fopen(argv[3], "wb");
		if (!f) { std::perror(argv[3]); return 2; }
		for (std::size_t i = 0; i < frames; ++i) std::fwrite(list[i].data(), sizeof(float), list[i].size(), f);
		std::fclose(f);
		std::printf("frames=%zu params=%zu audio=%zu\n", frames, frames ? list[0].size() : 0, audio.size());
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 1;
	}
	return 0;
}
