// Drop-in for the reference's OceanCurrents filter (src/filter/OceanCurrents.h:24-54): same place in the
// operator surface (an AsyncSubFilter created by OceanCurrentsMenu::makeFilter), compute through the C ABI.
#pragma once
#include <string>
#include "Filter.h"
#include "../../include/demiurge_hip.h"

class HipOceanCurrents : public AsyncSubFilter {
public:
    // pressure = OceanCurrentsMenu::pressure (reference OceanCurrents.h:21); n_outer replaces the reference's
    // hard-coded 10000 / while(true) (OceanCurrents.cpp:155,212); the iteration counts are its literals (:50,:701)
    HipOceanCurrents(Project *p, float pressure, int n_outer = 10000, int jacobi_iters = 5000, int diffusion_iters = 50)
        : pressure_(pressure), n_outer_(n_outer), jacobi_iters_(jacobi_iters), diffusion_iters_(diffusion_iters) { (void)p; }
    const std::string &error() const { return error_; }

protected:
    void run() override;

private:
    float pressure_;
    int n_outer_, jacobi_iters_, diffusion_iters_;
    std::string error_;
};
