// Binds the engine's arithmetic backend to the HIP implementations of include/cognn_hip.h.
#include "backend.h"

extern "C" const cognn_backend* cognn_default_backend(void) {
    static const cognn_backend be = {
#define X(name) &name,
        COGNN_BACKEND_FUNCS(X)
#undef X
    };
    return &be;
}
