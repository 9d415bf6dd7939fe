// TEST INFRASTRUCTURE — not part of the product.
//
// Captures, from the REAL GamaTTS pipeline, the input and the output of
//   GS::VTMControlModel::EventList::generateOutput()   (vtm_control_model/EventList.cpp:930-1091)
// for one text: the event list the text parser + rules produced (times, parameter targets,
// special-parameter targets, macro-intonation polynomials), the flags and pitch constants it runs
// with, and the float32 parameter frames it emits.  generateOutput() is then called again on the same
// list with other intonation settings (the drift generator's state carries over from call to call,
// exactly as in a Controller that synthesizes several chunks), each call being recorded.
// Our own code, compiled against the reference where it lies (oracle/Makefile target ref_full).
//
// usage: ref_tracks_capture <voice_data_dir> "<single sentence>" <out.bin>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "Controller.h"
#include "Index.h"
#include "Model.h"
#include "TextParser.h"

namespace {

void put_i32(FILE* f, std::int32_t v) { std::fwrite(&v, sizeof(v), 1, f); }
void put_f64(FILE* f, double v) { std::fwrite(&v, sizeof(v), 1, f); }

void dump_call(FILE* f, GS::VTMControlModel::Controller& c, const std::vector<std::vector<float>>& frames)
{
	auto& ev = c.eventList();
	const auto& cfg = c.vtmControlModelConfiguration();
	put_i32(f, static_cast<std::int32_t>(cfg.controlPeriod));
	put_i32(f, ev.macroIntonation());
	put_i32(f, ev.microIntonation());
	put_i32(f, ev.intonationDrift());
	put_i32(f, ev.smoothIntonation());
	put_f64(f, ev.initialPitch());
	put_f64(f, ev.meanPitch());
	put_f64(f, cfg.driftDeviation);
	put_f64(f, cfg.controlRate);
	put_f64(f, cfg.driftLowpassCutoff);
	const auto& list = ev.list();
	put_i32(f, static_cast<std::int32_t>(list.size()));
	for (const auto& e : list) {
		put_f64(f, e->time);
		put_f64(f, e->interpData ? 1.0 : 0.0);
		put_f64(f, e->interpData ? e->interpData->a : 0.0);
		put_f64(f, e->interpData ? e->interpData->b : 0.0);
		put_f64(f, e->interpData ? e->interpData->c : 0.0);
		put_f64(f, e->interpData ? e->interpData->d : 0.0);
		for (int j = 0; j < 16; ++j) put_f64(f, e->getParameter(j, false));
		for (int j = 0; j < 16; ++j) put_f64(f, e->getParameter(j, true));
	}
	put_i32(f, static_cast<std::int32_t>(frames.size()));
	for (const auto& fr : frames) std::fwrite(fr.data(), sizeof(float), 16, f);
}

} // namespace

int main(int argc, char** argv)
{
	if (argc != 4) {
		std::fprintf(stderr, "usage: %s voice_dir text out.bin\n", argv[0]);
		return 2;
	}
	try {
		const GS::Index index{argv[1]};
		auto model = std::make_unique<GS::VTMControlModel::Model>();
		model->load(index);
		if (model->parameterList().size() != 16) { std::fprintf(stderr, "expected 16 parameters\n"); return 2; }
		auto controller = std::make_unique<GS::VTMControlModel::Controller>(index, *model);
		auto parser = GS::TextParser::TextParser::getInstance(index, controller->vtmControlModelConfiguration().phoStrFormat);
		const std::string pho = parser->parse(argv[2]);
		std::vector<float> audio;
		controller->synthesizePhoneticStringToBuffer(pho, nullptr, audio);
		// call 0: what the pipeline itself produced (fresh drift generator, the voice's own flags)
		std::vector<std::vector<float>> first = controller->vtmParameterList();
		if (!first.empty()) first.pop_back(); // Controller::synthesize appended a copy of the last frame (Controller.cpp:283)
		auto& ev = controller->eventList();

		FILE* f = std::fopen(argv[3], "wb");
		if (!f) { std::perror(argv[3]); return 2; }
		std::fwrite("GVTR", 1, 4, f);
		put_i32(f, 1);
		const int n_calls = 6;
		put_i32(f, n_calls);
		dump_call(f, *controller, first);
		// further calls on the same list; the drift generator keeps running
		struct Setting { bool macro, micro, drift, smooth, reprepare; };
		const Setting settings[n_calls - 1] = {
			{true, true, true, true, false},    // same flags again: only the drift continuation differs
			{true, true, false, true, false},   // no drift
			{false, true, true, true, false},   // micro intonation + drift only
			{true, false, false, true, false},  // macro only
			{true, true, true, false, true},    // straight-line macro intonation (polynomials re-prepared)
		};
		for (const Setting& s : settings) {
			ev.setMacroIntonation(s.macro);
			ev.setMicroIntonation(s.micro);
			ev.setIntonationDrift(s.drift);
			ev.setSmoothIntonation(s.smooth);
			if (s.reprepare) {
				ev.clearMacroIntonation();
				ev.prepareMacroIntonationInterpolation();
			}
			std::vector<std::vector<float>> frames;
			ev.generateOutput(frames);
			dump_call(f, *controller, frames);
		}
		std::fclose(f);
		std::printf("events=%zu frames=%zu\n", ev.list().size(), first.size());
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 1;
	}
	return 0;
}
