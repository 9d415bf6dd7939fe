// gama_vtm_batch — batched form of `gama_tts vtm` (gama_tts/src/main.cpp:286-337):
//
//   gama_vtm_batch [-d device[,device...] | -d all] [-m | -f] <voice_dir> <out_dir> <param_file>...
//
// Every parameter file holds one utterance (one 16-float frame per line, the format written
// by `gama_tts tts -p`); all of them are synthesized in one launch per device (contiguous shards of the
// file list when several devices are named) and written as <out_dir>/<basename>.wav, scaled like
// Controller::writeOutputToFile.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <string>
#include <vector>

#include "batch_controller.hpp"

int main(int argc, char** argv)
{
	std::vector<int> devices{0};
	int precision = GVTM_PRECISION_F64;
	int i = 1;
	for (; i < argc && argv[i][0] == '-'; ++i) {
		if (std::strcmp(argv[i], "-d") == 0 && i + 1 < argc) {
			const std::string list = argv[++i];
			devices.clear();
			if (list == "all") {
				for (int d = 0; d < gvtm_device_count(); ++d) devices.push_back(d);
			} else {
				std::size_t pos = 0;
				while (pos <= list.size()) {
					const std::size_t comma = list.find(',', pos);
					devices.push_back(std::atoi(list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos).c_str()));
					if (comma == std::string::npos) break;
					pos = comma + 1;
				}
			}
			if (devices.empty()) { std::cerr << "no device" << std::endl; return EXIT_FAILURE; }
		}
		else if (std::strcmp(argv[i], "-m") == 0) precision = GVTM_PRECISION_MIXED;
		else if (std::strcmp(argv[i], "-f") == 0) precision = GVTM_PRECISION_F32; // reference model 1 (float) semantics
		else { std::cerr << "unknown option " << argv[i] << std::endl; return EXIT_FAILURE; }
	}
	if (argc - i < 3) {
		std::cerr << "usage: " << argv[0] << " [-d device[,device...] | -d all] [-m | -f] voice_dir out_dir param_file..." << std::endl;
		return EXIT_FAILURE;
	}
	try {
		gvtm::BatchController controller(argv[i], devices, precision);
		const std::string out_dir = argv[i + 1];
		// output names first: two inputs with the same basename would write the same WAV
		std::vector<std::string> names;
		std::set<std::string> seen;
		for (int a = i + 2; a < argc; ++a) {
			std::string base = argv[a];
			const auto slash = base.find_last_of('/');
			if (slash != std::string::npos) base = base.substr(slash + 1);
			const auto dot = base.find_last_of('.');
			if (dot != std::string::npos) base = base.substr(0, dot);
			if (!seen.insert(base).second) {
				std::cerr << "Two parameter files share the basename '" << base << "': their output would be the same file " << out_dir << '/' << base << ".wav." << std::endl;
				return EXIT_FAILURE;
			}
			names.push_back(base);
		}
		for (int a = i + 2; a < argc; ++a) {
			std::ifstream in(argv[a], std::ios_base::binary);
			if (!in) { std::cerr << "Could not open the file " << argv[a] << '.' << std::endl; return EXIT_FAILURE; }
			controller.addUtteranceFromStream(in);
		}
		controller.synthesize(gvtm::BatchController::Output::Pcm16); // the WAV files' own 16-bit samples come back from the device
		for (std::size_t u = 0; u < controller.size(); ++u) {
			const std::string path = out_dir + "/" + names[u] + ".wav";
			controller.writeWav(u, path);
			std::printf("%s: %zu samples, scale %.6g\n", path.c_str(), controller.sampleCount(u), controller.outputScale(u));
		}
	} catch (const std::exception& e) {
		std::cerr << "Exception: " << e.what() << std::endl;
		return EXIT_FAILURE;
	}
	return EXIT_SUCCESS;
}
