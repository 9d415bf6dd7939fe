/*
 * gama_vtm.h — C ABI of the MI355X-native batched vocal-tract-model synthesizer.
 *
 * This is the drop-in boundary for GamaTTS's VTM hot path.  Plain C: pointers,
 * sizes and error codes only (no C++/torch types).  The library behind it
 * (libgama_vtm.so) is HIP for gfx950; there is no CPU fallback — every entry
 * point that needs the device fails with GVTM_ERR_NO_DEVICE / GVTM_ERR_HIP
 * when it is missing.
 *
 * Reference interfaces each entry replaces (paths under gama_tts/src/):
 *
 *   gvtm_plan_create            VocalTractModel0/2/4 constructor: loadConfiguration +
 *                               initializeSynthesizer (vtm/VocalTractModel0.h:255-305, :338-392;
 *                               vtm/VocalTractModel2.h:322-376, :413-467; vtm/VocalTractModel4.h:370-515)
 *                               for a whole batch
 *   gvtm_plan_create_model5     VocalTractModel5 constructor (vtm/VocalTractModel5.h:375-421, :455-521); synthesis
 *                               then replaces its execSynthesisStep / vocalTract (:523-579, :632-730)
 *   gvtm_plan_info              VocalTractModel::internalSampleRate/outputSampleRate
 *                               (vtm/VocalTractModel.h:51-52) and the controlSteps of
 *                               Controller::synthesize (vtm_control_model/Controller.cpp:286)
 *   gvtm_output_count           outputBuffer().size() after finishSynthesis()
 *                               (vtm/VocalTractModel.h:58, vtm/SampleRateConverter.h:462-471)
 *   gvtm_synthesize_batch_*     Controller::synthesize + VocalTractModel::setAllParameters /
 *                               execSynthesisStep / finishSynthesis for B utterances
 *                               (vtm_control_model/Controller.cpp:277-313,
 *                               vtm/VocalTractModel0.h:396-445, :698-723)
 *   gvtm_stream_*               the same three calls as a STATEFUL object: what VocalTractModel keeps between
 *                               execSynthesisStep() calls (vtm/VocalTractModel0.h:221-252, :396-445), reset() (:309-326)
 *                               and finishSynthesis() (:720-723), so that an utterance can be handed over in pieces
 *                               (long tracks; the editor's interactive polling, InteractiveAudio.cpp:141-185)
 *   gvtm_synthesize_batch_host_pcm16
 *                               the path as `gama_tts vtm` ends it: Controller::synthesizeToFile's int16 samples
 *                               (Controller.cpp:236-252, :325-340; WAVEFileWriter.cpp:122-125)
 *   gvtm_synthesize_events_device
 *                               EventList::generateOutput (vtm_control_model/EventList.cpp:930-1091) followed by
 *                               Controller::synthesize in one call: event lists in, samples out
 *   gvtm_normalize_batch_device Controller::writeOutputToBuffer / writeOutputToFile scaling,
 *                               Util::calculateOutputScale (Controller.cpp:315-340,
 *                               vtm/VTMUtil.cpp:48-67, WAVEFileWriter.cpp:122-125)
 *
 * The GamaTTS plugin entry points GAMA_TTS_construct_vocal_tract_model /
 * GAMA_TTS_destruct_vocal_tract_model (vtm/VocalTractModelPlugin.cpp:40-50) are
 * exported by libgama_vtm_plugin.so, which is a thin C++ shim over this ABI
 * (see include/gama_vtm_plugin.h and INTEGRATION.md).
 */
#ifndef GAMA_VTM_H_
#define GAMA_VTM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the libraries are built with -fvisibility=hidden: these entry points are their exports */
#endif

#define GVTM_N_PARAM 16 /* pitch, glotVol, aspVol, fricVol, fricPos, fricCF, fricBW, r1..r8, velum
                           (vtm/VocalTractModel0.h:160-178) */

typedef enum gvtm_status {
	GVTM_OK = 0,
	GVTM_ERR_INVALID_ARGUMENT = 1, /* null pointer, bad size, bad configuration value */
	GVTM_ERR_NO_DEVICE = 2,        /* no HIP device / device index out of range */
	GVTM_ERR_HIP = 3,              /* a HIP runtime call failed (see gvtm_last_error) */
	GVTM_ERR_UNSUPPORTED = 4,      /* valid for the reference but not implemented on the device (e.g. a workgroup shape
	                                  whose buffers do not fit LDS) */
	GVTM_ERR_OUT_OF_MEMORY = 5
} gvtm_status;

/* Arithmetic the device path computes in. */
typedef enum gvtm_precision {
	GVTM_PRECISION_F64 = 0,  /* everything in fp64, like VocalTractModel0<double> */
	GVTM_PRECISION_MIXED = 1, /* fp64 sources, tube and filters; fp32 sample-rate converter (tables, window, MACs) */
	GVTM_PRECISION_F32 = 2    /* everything in fp32, like VocalTractModel0<float> (reference model 1) and
	                             VocalTractModel2<float,D>: design tables and constants computed in float too, every
	                             rounding reproduced — the output is bit-identical to the reference's FLOAT model
	                             (which itself differs from its double model by several percent of peak after a
	                             few seconds: oscillator phase, 47 instead of 49 FIR taps) */
} gvtm_precision;

/* Tube topology. */
typedef enum gvtm_tube_layout {
	GVTM_TUBE_10_6 = 0,  /* 10 oropharynx + 6 nasal sections: VocalTractModel0 / VocalTractModel2 (models 0, 2, 3) */
	GVTM_TUBE_30_18 = 1  /* 30 + 18 sections, scattering at region boundaries only: VocalTractModel4 (model 4,
	                        vtm/VocalTractModel4.h); SectionDelay 1 */
} gvtm_tube_layout;

/* The configuration keys VocalTractModel0/2::loadConfiguration reads from the merged
 * vtm.txt + variant file (vtm/VocalTractModel0.h:266-305), as numbers. */
typedef struct gvtm_config {
	double output_rate;
	int32_t waveform; /* 0 pulse, 1 sine */
	int32_t noise_modulation;
	double glottal_pulse_tp;
	double glottal_pulse_tn_min;
	double glottal_pulse_tn_max;
	double breathiness;
	double vocal_tract_length_offset;
	double vocal_tract_length;
	double temperature;
	double loss_factor;
	double mouth_coefficient;
	double nose_coefficient;
	double throat_cutoff;
	double throat_volume;
	double mix_offset;
	double global_radius_coef;
	double global_nasal_radius_coef;
	double aperture_radius;
	double nasal_radius[5]; /* nasal_radius_1 .. nasal_radius_5 */
	double radius_coef[8];  /* radius_1_coef .. radius_8_coef */
	int32_t section_delay;  /* VocalTractModel2's SectionDelay; 1 == VocalTractModel0 (models 0/2), 3 == model 3 */
	int32_t precision;      /* gvtm_precision */
	int32_t tube_layout;    /* gvtm_tube_layout */
	int32_t reserved_;      /* must be 0 */
} gvtm_config;

typedef struct gvtm_info {
	int32_t internal_sample_rate; /* Hz, vtm/VocalTractModel0.h:344 */
	uint32_t control_steps;       /* internal steps per control frame, Controller.cpp:286 */
	double output_rate;
	double control_rate;
	int32_t fir_taps;                 /* glottal-source FIR, WavetableGlottalSourceFIRFilter.h:86 */
	uint32_t time_register_increment; /* SampleRateConverter.h:145 */
	uint32_t phase_increment;         /* SampleRateConverter.h:154 (0 when up-sampling) */
	int32_t pad_size;                 /* SampleRateConverter.h:158-160 */
	int32_t upsampling;
	int32_t device;
	int32_t precision;
	int32_t section_delay;
	int32_t model5;                   /* 1 for a plan made by gvtm_plan_create_model5 */
	int32_t reserved_;
	double internal_rate_hz;          /* the internal rate as the model holds it: an integer for models 0-4, not for
	                                     model 5 (vtm/VocalTractModel5.h:465 keeps it in TFloat) */
} gvtm_info;

/* The configuration keys VocalTractModel5::loadConfiguration reads (vtm/VocalTractModel5.h:375-421), as numbers:
 * reference model 5 — 30 + 21 sections with flow junctions, Rosenberg B glottal source, pole-zero radiation
 * impedance at mouth and nose, Butterworth noise/source filters, glottal loss, differentiated output. */
typedef struct gvtm5_config {
	double output_rate;
	int32_t waveform;                         /* 0 pulse, 1 sine */
	int32_t noise_modulation;
	int32_t bypass;                           /* 1: the glottal signal goes straight to the resampler (:566-568) */
	int32_t constant_radius_mouth_impedance;  /* 0: the mouth impedance follows r8 every step */
	double glottal_pulse_tp;
	double glottal_pulse_tn_min;
	double glottal_pulse_tn_max;
	double breathiness;
	double vocal_tract_length_offset;
	double vocal_tract_length;
	double temperature;
	double loss_factor;
	double mix_offset;
	double global_radius_coef;
	double global_nasal_radius_coef;
	double nasal_radius[6];                   /* nasal_radius_2 .. nasal_radius_7 */
	double radius_coef[8];
	double glottal_noise_cutoff;
	double frication_noise_cutoff;
	double frication_factor;
	double min_glottal_loss;
	double max_glottal_loss;
	double glottal_lowpass_cutoff;
	double mouth_impedance_radius;            /* used when constant_radius_mouth_impedance != 0 */
	int32_t precision;                        /* GVTM_PRECISION_F64 only: the factory's model 5 is VocalTractModel5<double,1>
	                                             (vtm/VocalTractModel.cpp:47-48) */
	int32_t reserved_;                        /* must be 0 */
} gvtm5_config;

typedef struct gvtm_plan gvtm_plan;

/* device index for a design-only plan: tables, info and output counts are available on a
 * machine without a GPU; every synthesis entry point returns GVTM_ERR_NO_DEVICE. */
#define GVTM_DEVICE_NONE (-1)

/* Design-time tables of a plan, for inspection/tests. */
typedef enum gvtm_table {
	GVTM_TABLE_FIR = 0,       /* fir_taps doubles */
	GVTM_TABLE_SRC_H = 1,     /* 3328 doubles */
	GVTM_TABLE_SRC_DH = 2,    /* 3328 doubles */
	GVTM_TABLE_WAVETABLE = 3  /* 512 doubles */
} gvtm_table;

const char* gvtm_status_string(int status);
/* Message of the last failing call on this thread ("" if none). */
const char* gvtm_last_error(void);
/* Number of HIP devices visible (0 when there is none or the runtime is unusable). */
int gvtm_device_count(void);

/* Validates the configuration, designs the tables on the host (fp64) and uploads them to
 * `device`.  control_rate is 1000 / control_period Hz (VTMControlModelConfiguration.cpp:41). */
int gvtm_plan_create(const gvtm_config* config, double control_rate, int device, gvtm_plan** plan_out);
/* The same for reference model 5 (VocalTractModel5 constructor: loadConfiguration + initializeSynthesizer,
 * vtm/VocalTractModel5.h:375-421, :455-521).  The plan is used with the same synthesis entry points;
 * gvtm_plan_table() serves the resampler tables only.  Refused (GVTM_ERR_INVALID_ARGUMENT), as the reference's constructors
 * refuse them: an internal rate below 50 kHz (vocal tract longer than ~21 cm, PoleZeroRadiationImpedance.h:116-119),
 * glottal pulse timings outside RosenbergBGlottalSource's checks, Butterworth cutoffs outside 1 Hz .. 0.48 of the internal
 * rate; and, a limit of this implementation, an output rate above 3x the internal rate. */
int gvtm_plan_create_model5(const gvtm5_config* config, double control_rate, int device, gvtm_plan** plan_out);
void gvtm_plan_destroy(gvtm_plan* plan);
int gvtm_plan_info(const gvtm_plan* plan, gvtm_info* info_out);
/* Copies a design table into out[capacity]; returns the element count or a negative status. */
int gvtm_plan_table(const gvtm_plan* plan, int which, double* out, size_t capacity);

/* Samples finishSynthesis() leaves for an utterance of n_frames frames; (size_t)-1 for a null plan.  Includes the
 * reference converter's flush overrun (vtm/SampleRateConverter.h:298-308 with :462-471): on down-sampling plans
 * (reference models 3, 4, 5 at 44.1 / 48 kHz) about 0.4 % of the frame counts make flushBuffer()'s final dataEmpty()
 * convert one more lap of the 1024-sample ring (~750 extra samples computed from ring leftovers); the device
 * reproduces those samples, so such a count is LARGER than that of the next longer utterance. */
size_t gvtm_output_count(const gvtm_plan* plan, size_t n_frames);
/* max over n <= max_frames of gvtm_output_count(plan, n): the audio_stride that holds every utterance of a ragged
 * batch.  Equal to gvtm_output_count(plan, max_frames) on up-sampling plans. */
size_t gvtm_output_capacity(const gvtm_plan* plan, size_t max_frames);

/*
 * Batch synthesis, everything resident in device memory.
 *   d_params       [batch][max_frames][16] float32, reference frame order
 *   d_frame_counts [batch] int32 frames per utterance (<= max_frames), or NULL: all max_frames
 *   d_audio        [batch][audio_stride] float32, unscaled samples as in outputBuffer();
 *                  audio_stride >= gvtm_output_count(plan, max_frames) is required, gvtm_output_capacity(plan,
 *                  max_frames) holds every utterance of a ragged batch (samples beyond audio_stride are dropped,
 *                  d_out_counts still reports them); [count, audio_stride) of a row is left untouched
 *   d_out_counts   [batch] int64 samples per utterance, may be NULL
 *   d_maxabs       [batch] float32 max|x| per utterance, may be NULL
 *   hip_stream     hipStream_t (NULL = default stream); the call only enqueues work
 *                  (GVTM_PRECISION_F32 plans: the first call, and a later one with more frames than any before, also builds
 *                  and uploads the plan's noise-sample table for max_frames frames, 4 bytes per internal step, synchronously)
 */
int gvtm_synthesize_batch_device(gvtm_plan* plan, const float* d_params, const int32_t* d_frame_counts,
		size_t batch, size_t max_frames, float* d_audio, size_t audio_stride,
		int64_t* d_out_counts, float* d_maxabs, void* hip_stream);

/* Same with host buffers (H2D, kernel, D2H, synchronous).  frame_counts may be NULL.  A frame_counts[b] outside
 * [0, max_frames] fails that utterance only: out_counts[b] = -1, its row zeroed, the call still returns GVTM_OK
 * (gvtm_last_error() names the cause).  Rows are zero beyond their sample count.  The caller's current HIP device
 * is restored before returning (every entry point does). */
int gvtm_synthesize_batch_host(gvtm_plan* plan, const float* params, const int32_t* frame_counts,
		size_t batch, size_t max_frames, float* audio, size_t audio_stride,
		int64_t* out_counts, float* maxabs);

/* The same path ending where the reference's file writer ends (Controller::writeOutputToFile, Controller.cpp:325-340 with
 * WAVEFileWriter::writeSample, WAVEFileWriter.cpp:122-125): pcm[b][i] = round(x[b][i] * (0.95 / max|x[b]|) * 32767) as
 * int16, exactly the samples `gama_tts vtm` puts into its WAV file -- 2 bytes per sample cross PCIe instead of 4.
 *   pcm        [batch][pcm_stride] int16, rows zero beyond their sample count
 *   scales     [batch] the factor 0.95 / max|x| applied to each utterance (0 for silence), may be NULL
 *   maxabs     [batch] max|x| of the unscaled samples, may be NULL
 * Both host entries cut a batch of two or more machine-fulls (utterances per workgroup x compute units) into slices and
 * run   H2D frames(i+1) || kernel(i) [+ scaling(i)] || D2H samples(i-1)   on three streams.  The overlap is real when
 * the host buffers are page-locked (gvtm_host_alloc, hipHostMalloc, hipHostRegister); pageable buffers give the same
 * bytes, with the runtime staging the copies. */
int gvtm_synthesize_batch_host_pcm16(gvtm_plan* plan, const float* params, const int32_t* frame_counts,
		size_t batch, size_t max_frames, int16_t* pcm, size_t pcm_stride,
		int64_t* out_counts, float* maxabs, float* scales);

/* Page-locked host memory for the buffers of the host entries (hipHostMalloc, portable across devices), for callers
 * that do not link the HIP runtime themselves.  gvtm_host_free(NULL) is a no-op. */
int gvtm_host_alloc(size_t bytes, void** ptr_out);
void gvtm_host_free(void* ptr);

/* ---------------------------------------------------------------------------------------------
 * Streams: a batch of utterances synthesized piece by piece.
 *
 * The reference model is an object with state: every execSynthesisStep() continues where the last one stopped, and
 * the caller may read outputBuffer() whenever it likes (the editor does, after every step).  A gvtm_stream keeps that
 * state for `batch` independent utterances in device memory between launches: section delay lines, filter memories,
 * oscillator phase, noise seed, the decimator's and the converter's buffers and the converter's position.  Pushing an
 * utterance in any number of pieces and finishing it yields exactly the samples of the one-shot entry points
 * (bit for bit, in every precision).
 *
 *   gvtm_stream_push    appends frames; synthesizes every frame whose successor is known (the driver loop interpolates
 *                       each frame TOWARDS the next one, Controller.cpp:297-300), in multiples of a few frames (the
 *                       wavefronts' recurrences are unrolled: 12 internal steps), keeps the rest; returns the new samples
 *   gvtm_stream_finish  synthesizes what is kept (the last frame stands for its own successor, Controller.cpp:283) and
 *                       flushes the converter (finishSynthesis()); maxabs = max |sample| of the whole utterance
 *   gvtm_stream_reset   every utterance back to the state after construction (VocalTractModel::reset())
 *
 * Utterances pushed in lockstep (same frame counts every time) share workgroups like a one-shot batch; otherwise a
 * workgroup takes one utterance.  Plans of reference model 5 have streams too (vtm/VocalTractModel5.h:523-579: its scans,
 * filter memories, section flows, radiation-impedance memories, converter ring and the difference filter's look-back are
 * the state; pushes go in multiples of four internal steps).
 */
typedef struct gvtm_stream gvtm_stream;

int gvtm_stream_create(gvtm_plan* plan, size_t batch, gvtm_stream** stream_out);
void gvtm_stream_destroy(gvtm_stream* stream);
int gvtm_stream_reset(gvtm_stream* stream);
/* Samples per utterance that a push of at most max_new_frames frames, or the finish after it, can return: the
 * audio_stride to allocate. */
size_t gvtm_stream_capacity(const gvtm_stream* stream, size_t max_new_frames);
/* params [batch][max_frames][16] float32 (host), frame_counts [batch] new frames per utterance or NULL (max_frames each);
 * audio [batch][audio_stride] float32 (host) receives the new samples of each utterance from index 0, out_counts[b]
 * how many (may be NULL); rows are zero beyond their count.  Synchronous. */
int gvtm_stream_push(gvtm_stream* stream, const float* params, const int32_t* frame_counts, size_t max_frames,
		float* audio, size_t audio_stride, int64_t* out_counts);
int gvtm_stream_finish(gvtm_stream* stream, float* audio, size_t audio_stride, int64_t* out_counts, float* maxabs);

/* Output scaling of Controller::writeOutputToBuffer / writeOutputToFile: scale = 0.95 / max|x|
 * (0 when max < 1e-30).  Exactly one of d_out_f32 / d_out_i16 may be non-NULL; i16 applies the
 * WAVEFileWriter rounding round(x * 32767). d_counts may be NULL (then audio_stride samples). */
int gvtm_normalize_batch_device(gvtm_plan* plan, const float* d_audio, size_t batch, size_t audio_stride,
		const int64_t* d_counts, const float* d_maxabs, float* d_out_f32, int16_t* d_out_i16,
		float* d_scales, void* hip_stream);

/* Average device time (ms) of the dominant synthesis kernel over the launches since the last
 * call, measured with HIP events on the launch stream; resets the accumulator.  Timing must
 * have been enabled with gvtm_plan_set_timing(plan, 1).  Returns < 0 when nothing was timed. */
int gvtm_plan_set_timing(gvtm_plan* plan, int enabled);
double gvtm_plan_take_kernel_ms(gvtm_plan* plan, int* launches_out);

/* ---------------------------------------------------------------------------------------------
 * Parameter-track generation: the step in front of the vocal-tract path, for a batch.
 * Replaces EventList::generateOutput() (vtm_control_model/EventList.cpp:930-1091) together with
 * DriftGenerator::drift() (vtm_control_model/DriftGenerator.cpp:72-84): event lists in, one
 * float32[16] frame per control period out, laid out as gvtm_synthesize_batch_device() reads them
 * (the frames never leave the device).  Bit-identical to the reference.
 */

/* One EventList event (vtm_control_model/EventList.h:117-160), flattened.  A parameter the event does
 * not set holds Event::EMPTY_PARAMETER = +infinity (HUGE_VAL). */
typedef struct gvtm_event {
	int32_t time_ms;     /* Event::time */
	int32_t has_interp;  /* Event::interpData present */
	double interp[4];    /* InterpolationData a, b, c, d (macro intonation, EventList.h:105-115) */
	double param[16];    /* Event::parameters */
	double special[16];  /* Event::specialParameters */
} gvtm_event;

typedef struct gvtm_track_config {
	int32_t control_period_ms;   /* EventList::controlPeriod_ (1..4), = 1000 / control rate */
	int32_t macro_intonation;    /* EventList flags (EventList.h:182-192) */
	int32_t micro_intonation;
	int32_t intonation_drift;
	int32_t smooth_intonation;
	int32_t reserved_;           /* must be 0 */
	double initial_pitch;        /* EventList::initialPitch_ (Controller.cpp:70) */
	double mean_pitch;           /* EventList::meanPitch_ = pitch_offset + reference_glottal_pitch (Controller.cpp:71) */
	double drift_deviation;      /* DriftGenerator::setUp(deviation, sampleRate, lowpassCutoff), Controller.cpp:73 */
	double drift_sample_rate;
	double drift_lowpass_cutoff;
} gvtm_track_config;

/* DriftGenerator state (noise seed, Butterworth filter memory).  The reference keeps one generator per
 * Controller, running on from chunk to chunk; a fresh one is {0.7892347, 0, 0, 0, 0}. */
typedef struct gvtm_drift_state {
	double seed, x1, x2, y1, y2;
} gvtm_drift_state;

/* Frames generateOutput() pushes for one event list (host-side, no device needed); (size_t)-1 on a
 * bad configuration. */
size_t gvtm_tracks_frame_count(const gvtm_track_config* config, const gvtm_event* events, size_t n_events);

/*
 * Batch generation, everything resident in device memory.
 *   d_events        all utterances' events back to back
 *   d_event_offsets [batch + 1] int64: utterance b owns events [offsets[b], offsets[b+1])
 *   d_params        [batch][max_frames][16] float32 out; frames beyond max_frames are dropped
 *   d_frame_counts  [batch] int32 out: frames generateOutput() produces (may exceed max_frames), may be NULL
 *   d_drift         [batch] in/out drift-generator states, or NULL: a fresh generator per utterance
 * d_params / d_frame_counts are exactly what gvtm_synthesize_batch_device() takes.
 */
int gvtm_generate_tracks_device(int device, const gvtm_track_config* config, const gvtm_event* d_events,
		const int64_t* d_event_offsets, size_t batch, size_t max_frames, float* d_params, int32_t* d_frame_counts,
		gvtm_drift_state* d_drift, void* hip_stream);

/*
 * Event lists in, audio out, in one call: gvtm_generate_tracks_device into a frame buffer the plan owns, then
 * gvtm_synthesize_batch_device, both enqueued on hip_stream (the frames never leave the device and the caller never sees
 * them).  Works for every plan, reference model 5 included.
 *   config          its control_period_ms must match the plan's control rate (1000 / control_rate)
 *   max_frames      frames per utterance the rows are sized for (gvtm_tracks_frame_count on the host); a list that yields
 *                   more is cut there
 *   d_frame_counts  [batch] int32 out: frames each list yields, may be NULL
 *   d_drift         [batch] in/out drift-generator states, or NULL (a fresh generator per utterance)
 * The remaining arguments are gvtm_synthesize_batch_device's.
 */
int gvtm_synthesize_events_device(gvtm_plan* plan, const gvtm_track_config* config, const gvtm_event* d_events,
		const int64_t* d_event_offsets, size_t batch, size_t max_frames, float* d_audio, size_t audio_stride,
		int32_t* d_frame_counts, int64_t* d_out_counts, float* d_maxabs, gvtm_drift_state* d_drift, void* hip_stream);

/* Same with host buffers (H2D, kernel, D2H, synchronous). */
int gvtm_generate_tracks_host(int device, const gvtm_track_config* config, const gvtm_event* events,
		const int64_t* event_offsets, size_t batch, size_t max_frames, float* params, int32_t* frame_counts,
		gvtm_drift_state* drift);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif

#endif /* GAMA_VTM_H_ */
