// Drop-in for the reference's OceanCurrents filter (src/filter/OceanCurrents.h:24-54): same place in the
// operator surface (an AsyncSubFilter created by OceanCurrentsMenu::makeFilter), compute through the C ABI.
#pragma once
#include <string>
#include <vector>
#include "Filter.h"
#include "../../include/demiurge_hip.h"

class HipOceanCurrents : public AsyncSubFilter {
public:
    ~HipOceanCurrents() override { stop(); }           // the worker is inside run(), reading the members below: joined before they go
    // pressure = OceanCurrentsMenu::pressure (reference OceanCurrents.h:21); n_outer replaces the reference's
    // hard-coded 10000 / while(true) (OceanCurrents.cpp:155,212); the iteration counts are its literals (:50,:701)
    // devices: the GPUs the solve is split over, one latitude band each ({0} = one GPU; {0,0} = two bands on GPU 0, a
    // rehearsal); ghost = halo rows per band side = sweeps between exchanges.  (Not a limit on the currents: an advect whose
    // back-trace -- |v| * 24 h, OceanCurrents.cpp:294-299, ~96 rows at H = 8192 once the wind has spun the ocean up -- is longer
    // than that gathers from a deeper copy of v that dmg_multi_step exchanges for that step, grown on demand.)
    HipOceanCurrents(Project *p, float pressure, int n_outer = 10000, int jacobi_iters = 5000, int diffusion_iters = 50,
                     std::vector<int> devices = { 0 }, int ghost = 32)
        : pressure_(pressure), n_outer_(n_outer), jacobi_iters_(jacobi_iters), diffusion_iters_(diffusion_iters),
          devices_(std::move(devices)), ghost_(ghost) { (void)p; }
    const std::string &error() const { return error_; }

protected:
    void run() override;

private:
    float pressure_;
    int n_outer_, jacobi_iters_, diffusion_iters_;
    std::vector<int> devices_;
    int ghost_;
    std::string error_;
};
