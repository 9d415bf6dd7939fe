/*
 * gama_vtm_plugin.h — the GamaTTS VocalTractModel plugin boundary served by
 * libgama_vtm_plugin.so.
 *
 * GamaTTS loads a vocal-tract model with `model = 2000` + `dll_path = <this .so>` in vtm.txt
 * (gama_tts/src/vtm/VocalTractModel.cpp:52-55, built with -DGAMATTS_ENABLE_VTM_PLUGINS=ON).
 * Its loader resolves exactly the two symbols below (vtm/VocalTractModelPlugin.cpp:40-50,
 * :76-85) and treats the returned pointer as a `GS::VTM::VocalTractModel*`
 * (vtm/VocalTractModelPlugin.cpp:87), i.e. the object must have the Itanium-ABI vtable
 *
 *     ~VocalTractModel() (two slots), reset(), internalSampleRate(), outputSampleRate(),
 *     setParameter(int, float), setAllParameters(const std::vector<float>&),
 *     execSynthesisStep(), finishSynthesis(), outputBuffer() -> std::vector<float>&
 *
 * in this order (vtm/VocalTractModel.h:46-59).  `config_data` is the host's
 * `const GS::ConfigurationData*` (two std::string members followed by a
 * std::unordered_map<std::string, std::string>, ConfigurationData.h:58-65); the plugin reads the
 * key/value map through a layout-compatible mirror and therefore has to be built with the
 * same libstdc++ (_GLIBCXX_USE_CXX11_ABI=1) as the host.
 *
 * Behaviour
 *   - Batch protocol (`is_interactive == 0`, Controller / `gama_tts tts|vtm` / speechd): parameters are recorded per
 *     execSynthesisStep() and the whole utterance is synthesized on the GPU inside finishSynthesis() (Controller reads
 *     outputBuffer() only after finishSynthesis(), Controller.cpp:226-235).  reset() and every finishSynthesis(),
 *     failed or not, leave a clean recording for the next utterance.
 *   - Interactive protocol (`is_interactive != 0`, the editor: gama_tts_editor/src/interactive/InteractiveAudio.cpp:
 *     141-185 polls outputBuffer() after every execSynthesisStep() until it holds enough samples, and empties it as it
 *     reads): the steps go to a gvtm_stream (include/gama_vtm.h) in blocks of `gpu_interactive_block` steps (default
 *     996; the reference's converter itself hands samples over every 998 steps, SampleRateConverter.h:277-281) and the
 *     new samples are appended to outputBuffer() after each block; finishSynthesis() synthesizes what is left and
 *     flushes; reset() starts over.  The sample stream is the batch protocol's, bit for bit
 *     (tests/test_gpu_dropin.py::test_plugin_interactive_protocol_through_reference_loader; with `gpu_model = 5`:
 *     ::test_plugin_model5_interactive_protocol_through_reference_loader).
 *   - No exception crosses the C boundary; any failure inside construct returns NULL and
 *     writes the reason to stderr.
 *   - Optional extra keys in vtm.txt: `gpu_device` (int, default 0), `gpu_precision`
 *     ("f64" default = model 0 / 2 / 3 / 4 semantics | "mixed" | "f32" = the float models, i.e. what
 *     `model = 1` selects in the reference), `section_delay` (1..4, default 1: VocalTractModel0 semantics;
 *     3 reproduces model 3), `tube_layout` (0 default; 1 = the 30+18-section tube of model 4),
 *     `gpu_model` (5 = the voice holds VocalTractModel5's keys and the plugin stands in for reference model 5),
 *     `gpu_interactive_block` (steps per launch in the interactive protocol, rounded up to a multiple of 12).
 */
#ifndef GAMA_VTM_PLUGIN_H_
#define GAMA_VTM_PLUGIN_H_

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the libraries are built with -fvisibility=hidden: these entry points are their exports */
#endif

void* GAMA_TTS_construct_vocal_tract_model(const void* config_data, int is_interactive);
void GAMA_TTS_destruct_vocal_tract_model(void* vtm);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif

#endif /* GAMA_VTM_PLUGIN_H_ */
