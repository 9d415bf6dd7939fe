// TEST INFRASTRUCTURE — not part of the product.
//
// The reference's `gama_tts vtm` (gama_tts/src/main.cpp:286-337) as a bare driver: parameter text file in, 16-bit WAV
// out through Controller::synthesizeToFile(std::istream&, const char*) (vtm_control_model/Controller.cpp:270-275), i.e.
// getParametersFromStream (:170-192) -> synthesize (:277-313) -> writeOutputToFile (:315-328) ->
// WAVEFileWriter (WAVEFileWriter.cpp:62-125).  Our own code; compiled against the reference where it lies
// (oracle/Makefile target ref_full); never shipped.  It pins SURVEY.md 8(a) row a16 (output scaling and int16 rounding)
// with bytes the reference itself wrote (tests/golden/make_wav_golden.py).
//
// usage: ref_wav_capture <voice_data_dir> <params.txt> <out.wav>
#include <cstdio>
#include <fstream>
#include <memory>

#include "Controller.h"
#include "Index.h"
#include "Model.h"

int main(int argc, char** argv)
{
	if (argc != 4) {
		std::fprintf(stderr, "usage: %s voice_dir params.txt out.wav\n", argv[0]);
		return 2;
	}
	try {
		std::ifstream in(argv[2], std::ios_base::binary);
		if (!in) { std::perror(argv[2]); return 2; }
		const GS::Index index{argv[1]};
		auto model = std::make_unique<GS::VTMControlModel::Model>();
		model->load(index);
		auto controller = std::make_unique<GS::VTMControlModel::Controller>(index, *model);
		controller->synthesizeToFile(in, argv[3]);
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 1;
	}
	return 0;
}
