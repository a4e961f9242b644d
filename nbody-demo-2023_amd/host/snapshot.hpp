// snapshot.hpp -- state snapshot I/O ("the step after the path").
// The reference keeps its state only in RAM and loses it at exit (ver7/GSimulation.cpp:265-279).
// Format (little endian): "NBXSNAP1", int32 n, int32 precision_bits, int64 steps_done, then the
// seven ParticleSoA arrays pos_x pos_y pos_z vel_x vel_y vel_z mass, n elements each.
#ifndef NBX_HOST_SNAPSHOT_HPP
#define NBX_HOST_SNAPSHOT_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "Particle.hpp"

namespace nbx_snapshot {

struct Header {
  char magic[8];
  int32_t n;
  int32_t precision;
  int64_t steps_done;
};

inline bool save(const std::string& path, const ParticleSoA* p, int n, long long steps_done, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) { *err = "cannot open " + path + " for writing"; return false; }
  Header h;
  std::memcpy(h.magic, "NBXSNAP1", 8);
  h.n = n; h.precision = 8 * (int)sizeof(real_type); h.steps_done = steps_done;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
  const real_type* arrays[7] = {p->pos_x, p->pos_y, p->pos_z, p->vel_x, p->vel_y, p->vel_z, p->mass};
  for (int k = 0; k < 7 && ok; ++k) ok = std::fwrite(arrays[k], sizeof(real_type), (size_t)n, f) == (size_t)n;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) *err = "short write to " + path;
  return ok;
}

inline bool read_header(const std::string& path, Header* h, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { *err = "cannot open " + path; return false; }
  bool ok = std::fread(h, sizeof *h, 1, f) == 1 && std::memcmp(h->magic, "NBXSNAP1", 8) == 0;
  std::fclose(f);
  if (!ok) *err = path + " is not an NBXSNAP1 snapshot";
  return ok;
}

inline bool load(const std::string& path, ParticleSoA* p, int n, long long* steps_done, std::string* err) {
  Header h;
  if (!read_header(path, &h, err)) return false;
  if (h.n != n) { *err = "snapshot holds " + std::to_string(h.n) + " bodies, the run has " + std::to_string(n); return false; }
  if (h.precision != 8 * (int)sizeof(real_type)) { *err = "snapshot precision does not match this executable"; return false; }
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { *err = "cannot open " + path; return false; }
  bool ok = std::fseek(f, (long)sizeof h, SEEK_SET) == 0;
  real_type* arrays[7] = {p->pos_x, p->pos_y, p->pos_z, p->vel_x, p->vel_y, p->vel_z, p->mass};
  for (int k = 0; k < 7 && ok; ++k) ok = std::fread(arrays[k], sizeof(real_type), (size_t)n, f) == (size_t)n;
  std::fclose(f);
  if (!ok) *err = "short read from " + path;
  *steps_done = h.steps_done;
  return ok;
}

}  // namespace nbx_snapshot
#endif
