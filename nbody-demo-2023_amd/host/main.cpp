// main.cpp -- `./nbody.x [nPart [nSteps]]`, the reference's command line (ver7/main.cpp:25-46):
// argv[1] -> number of particles; the step count is read only when EXACTLY two arguments are
// given (the reference tests argc==3, so a third argument silently disables it); no validation;
// exit status 0.
#include <cstdlib>

#include "GSimulation.hpp"

int main(int argc, char** argv) {
  GSimulation sim;
  if (argc > 1) {
    sim.set_number_of_particles(std::atoi(argv[1]));
    if (argc == 3) sim.set_number_of_steps(std::atoi(argv[2]));
  }
  sim.start();
  return 0;
}
