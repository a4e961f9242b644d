// main_ver5.cpp -- the ver5_all command line (ver5_all/main.cpp:23-66) in front of the same GSimulation:
//   ./nbody_v5.x [nPart [nSteps [cpu|gpu|cpu+gpu [cpu_ratio [dim0 dim1]]]]]
// Differences from ver7's main: nSteps is read whenever argc > 2, the device word is echoed, and the
// banner comes from main (compile GSimulation.cpp with -DNBX_BANNER_IN_MAIN).  There is no CPU
// engine behind libnbx: start() refuses "cpu" and runs "cpu+gpu" on the GPU alone (cpu_ratio is accepted
// and ignored); dim0 is the reference's block size (fixed at 256 here), dim1 selects bodies per lane.
// The reference's own ver5_all/main.cpp builds against this class unchanged (tests/test_reference_mains.py);
// this file differs from it only in reading argv[6] when it exists.
#include <cstdlib>
#include <iostream>
#include <string>

#include "GSimulation.hpp"

int main(int argc, char** argv) {
  GSimulation sim;
  if (argc > 1) {
    sim.set_number_of_particles(std::atoi(argv[1]));
    if (argc > 2) sim.set_number_of_steps(std::atoi(argv[2]));
    if (argc > 3) {
      const std::string dev = argv[3];
      std::cout << dev << std::endl;
      if (dev == "cpu") sim.set_devices(1);
      if (dev == "gpu") sim.set_devices(2);
      if (dev == "cpu+gpu") sim.set_devices(3);
    }
    if (argc > 4) sim.set_cpu_ratio((float)std::atof(argv[4]));
    if (argc > 5) sim.set_thread_dim0(std::atoi(argv[5]));
    if (argc > 6) sim.set_thread_dim1(std::atoi(argv[6]));
  }
  sim.init_mpi();
  if (sim.world_rank == 0) {
    std::cout << "===============================" << std::endl;
    std::cout << " Initialize Gravity Simulation" << std::endl;
  }
  sim.start();
  return 0;
}
