// Particle.hpp -- particle containers with the reference's member names
// (ver7/Particle.hpp:26-41 AoS `Particle`, :43-58 SoA `ParticleSoA`), so code written
// against the reference's GSimulation/Particle surface compiles unchanged.
// Only ParticleSoA is used by the hot path; Particle is kept for surface parity (ver0-ver2).
#ifndef NBX_HOST_PARTICLE_HPP
#define NBX_HOST_PARTICLE_HPP

#include <cstddef>

#include "types.hpp"

struct Particle {
  real_type pos[3];
  real_type vel[3];
  real_type acc[3];
  real_type mass;

  Particle() { init(); }
  void init() {
    for (int k = 0; k < 3; ++k) pos[k] = vel[k] = acc[k] = real_type(0);
    mass = real_type(0);
  }
};

struct ParticleSoA {
  real_type *pos_x, *pos_y, *pos_z;
  real_type *vel_x, *vel_y, *vel_z;
  real_type *acc_x, *acc_y, *acc_z;
  real_type *mass;

  ParticleSoA() { init(); }
  void init() {
    pos_x = pos_y = pos_z = NULL;
    vel_x = vel_y = vel_z = NULL;
    acc_x = acc_y = acc_z = NULL;
    mass = NULL;
  }
};

#endif
