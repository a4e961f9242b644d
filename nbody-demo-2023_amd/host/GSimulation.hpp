// GSimulation.hpp -- host mirror of the reference's simulation object.
//
// Same public surface as ver7/GSimulation.hpp:36-80 (ctor, dtor, init(),
// set_number_of_particles, set_number_of_steps, start()) plus the ver5_all knobs
// (ver5_all/GSimulation.hpp:51-58) so either main() links against it.  What differs is
// below start(): the per-step loops run on the MI355X through the C-ABI of include/nbx.h
// instead of OpenMP loops; allocation, initial conditions, timing and printing stay here.
#ifndef NBX_HOST_GSIMULATION_HPP
#define NBX_HOST_GSIMULATION_HPP

// the standard headers the reference's GSimulation.hpp provides to its includers (ver5_all/GSimulation.hpp:25-30;
// ver5_all/main.cpp uses std::cout and std::string without including anything else)
#include <iomanip>
#include <iostream>
#include <random>
#include <string>

#include "Particle.hpp"
#include "cpu_time.hpp"

struct nbx_ctx;

class GSimulation {
 public:
  GSimulation();
  ~GSimulation();

  void init();  // declared but never defined in the reference (ver7/GSimulation.hpp:42); here:
                // allocate the particle store and draw the seed-42 initial conditions
  void set_number_of_particles(int N);
  void set_number_of_steps(int N);
  void start();

  // ver5_all/GSimulation.hpp:51-58 -- accepted for CLI compatibility
  void set_cpu_ratio(const float& r) { _cpu_ratio = r; }
  void set_thread_dim0(const int& d) { _thread_dim0 = d; }
  void set_thread_dim1(const int& d) { _thread_dim1 = d; }
  int get_thread_dim0() { return _thread_dim0; }
  int get_thread_dim1() { return _thread_dim1; }
  int get_cpu_ratio() const { return (int)_cpu_ratio; }
  void set_devices(int N) { _devices = N; }  // 1 = "cpu" (refused by start(): no CPU engine), 2 = "gpu", 3 = "cpu+gpu"
  int get_devices() { return _devices; }

  // ver5_all/GSimulation.hpp:60-65 -- the MPI surface.  One process: rank 0 of 1 owning all bodies.  Several
  // processes (one per GPU; NBODY_WORLD / NBODY_RANK in the environment, or torchrun's WORLD_SIZE / RANK when
  // NBODY_USE_TORCHRUN_ENV=1 opts in): the
  // i-block partition of nbx_partition (include/nbx.h); see init_mpi() in GSimulation.cpp.  Only rank 0 prints.
  int world_rank;
  int world_size;
  int npp;          // bodies this process owns
  int* npp_global;  // [world_size] bodies per rank
  void init_mpi();

  // read-only views for embedding code and tests (the reference keeps these private)
  const ParticleSoA* particle_store() const { return particles; }
  real_type kinetic_energy() const { return _kenergy; }
  double total_time() const { return _totTime; }
  double total_flops() const { return _totFlops; }

 private:
  ParticleSoA* particles;

  int _npart;        // number of particles
  int _nsteps;       // number of integration steps
  real_type _tstep;  // time step
  int _sfreq;        // sample (print) frequency
  real_type _kenergy;
  double _totTime;
  double _totFlops;

  float _cpu_ratio;
  int _thread_dim0, _thread_dim1, _devices;

  bool _allocated;
  int _alloc_n;

  // one process per GPU (init_mpi): where rank 0 listens for the start-up rendezvous, and this process's GPU
  bool _multiprocess;
  std::string _master_addr;
  int _master_port, _local_rank;
  std::string _world_error;  // a bad NBODY_WORLD / NBODY_RANK, reported by init_mpi() / start() (never by the constructor)
  void read_world_env();
  void require_world() const;

  void allocate_store(int n);
  void release_store();

  void init_pos();
  void init_vel();
  void init_acc();
  void init_mass();

  void set_npart(const int& N) { _npart = N; }
  int get_npart() const { return _npart; }
  void set_tstep(const real_type& dt) { _tstep = dt; }
  real_type get_tstep() const { return _tstep; }
  void set_nsteps(const int& n) { _nsteps = n; }
  int get_nsteps() const { return _nsteps; }
  void set_sfreq(const int& sf) { _sfreq = sf; }
  int get_sfreq() const { return _sfreq; }

  void print_header();
};

#endif
