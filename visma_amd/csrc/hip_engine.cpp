// hip_engine.cpp -- the factory of the product engine (class HipEngine: hip_engine.hpp; its methods: hip_engine_clouds.cpp,
// hip_engine_passes.cpp, hip_engine_comm.cpp).
#include "hip_engine.hpp"

namespace visma {
namespace drv {
Engine *new_hip_engine(int device, int *rc, std::string *err)
{
    std::unique_ptr<HipEngine> e(new HipEngine(device));
    const int r = e->init();
    if (rc) *rc = r;
    if (r) { if (err) *err = e->error(); return nullptr; }
    return e.release();
}

}  // namespace drv
}  // namespace visma
