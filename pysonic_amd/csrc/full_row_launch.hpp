// pysonic_amd/csrc/full_row_launch.hpp -- what full_lib.hip needs of full_row_lib.hip
#pragma once
#include <vector>

#include "full_core.hpp"
#include "hybrid_core.hpp"

// neuron ids 2 .. 11: LTS, RE, TC, STN, IB and the data-driven HHseg, SWnode, MRGnode, SUseg, FHnode
bool full_row_available(int neuron_id);
// ... and whether THIS parameter block has a row layout (a data-driven block whose currents do not fit one quad of
// lanes each, or whose gates fall on the lanes of U, Z, ng, Qm, has none: the lane kernel runs it)
bool full_row_usable(int neuron_id, const std::vector<double> &params);
// whether the kernel that also holds RODAS4 exists for the neuron (full_row.hpp, RowModel::DEVICE_STIFF)
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
