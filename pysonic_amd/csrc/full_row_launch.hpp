// pysonic_amd/csrc/full_row_launch.hpp -- what full_lib.hip needs of full_row_lib.hip
#pragma once
#include <vector>

#include "full_core.hpp"

// LTS, RE, TC, STN, IB (neuron ids 2 .. 6)
bool full_row_available(int neuron_id);
// Launches the row-cooperative kernel of the detailed model on the null stream for the D.n configurations of D.
// *specs_out: a device allocation (lane descriptions) to hipFree once the kernel has ended.
int launch_full_row(int neuron_id, const sonic::FullDev &D, const sonic::BLSParams &p, const std::vector<double> &params,
                    int device, void **specs_out);
