// Parameter-track generation on the device: the step in front of the vocal-tract path.
// EventList::generateOutput (vtm_control_model/EventList.cpp:930-1091) for a batch of event lists.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/gama_vtm.h"
#include "vtm_design.hpp"

namespace gvtm {

struct TrackArgs {
	TrackConstants k;
	const gvtm_event* events;      // all utterances back to back
	const int64_t* event_offsets;  // [batch + 1]
	size_t batch;
	size_t max_frames;
	float* params;                 // [batch][max_frames][16]
	int32_t* frame_counts;         // [batch] or null
	gvtm_drift_state* drift;       // [batch] in/out, or null (fresh generator per utterance)
};

hipError_t launch_tracks(const TrackArgs& args, hipStream_t stream);

} // namespace gvtm
