// comm.h -- the handle's RCCL communicator (comm.cpp): the one exchange the reference's frame-parallel encoder has,
// ParSink's ordered gather (src/par.rs:67-95), for hosts that run one process per GPU.
#ifndef FLACENC_HIP_COMM_H_
#define FLACENC_HIP_COMM_H_

#include <string>

#include "../../include/flacenc_hip.h"

namespace flacenc_hip {

struct CommState;  // comm.cpp
// the handle's slot for it, its device and its error string (flacenc_hip_api.cpp)
CommState*& handle_comm_slot(flacenc_hip_handle* h);
int handle_device(const flacenc_hip_handle* h);
void handle_set_error(flacenc_hip_handle* h, const std::string& what);
// flacenc_hip_destroy: ncclCommDestroy + delete
void comm_release(CommState* c);

}  // namespace flacenc_hip

#endif
