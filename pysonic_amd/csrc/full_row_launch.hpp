// pysonic_amd/csrc/full_row_launch.hpp -- what full_lib.hip needs of full_row_lib.hip
#pragma once
#include <vector>

#include "full_core.hpp"
#include "hybrid_core.hpp"

// LTS, RE, TC, STN, IB (neuron ids 2 .. 6)
bool full_row_available(int neuron_id);
// whether the Rosenbrock variant of the row kernel exists for the neuron (not for TC: full_row.hpp, RowModel)
bool full_row_stiff_available(int neuron_id);
// Launches the row-cooperative kernel of the detailed model on the null stream for the D.n configurations of D
// (those D.sel lists, if set): the explicit 8(5,3) pair, or -- stiff -- RODAS4 from the start.
// *specs: in, null or the allocation a previous call returned; out, a device allocation (lane descriptions) to
// hipFree once the kernels have ended.
int launch_full_row(int neuron_id, const sonic::FullDev &D, const sonic::BLSParams &p, const std::vector<double> &params,
                    int device, bool stiff, void **specs);
// The row-cooperative kernel of the hybrid scheme (hybrid_row.hpp) for the D.n configurations of D; stiff: the build
// whose dense periods run on RODAS4 (full_row_stiff_available(neuron_id)); *specs as above.
int launch_hybrid_row(int neuron_id, const sonic::HybridDev &D, const sonic::BLSParams &p,
                      const std::vector<double> &params, int device, bool stiff, void **specs);
