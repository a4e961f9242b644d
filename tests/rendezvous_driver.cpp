// rendezvous_driver.cpp -- test harness for host/rendezvous.hpp (CPU only): one process = one rank.
//   rendezvous_driver <rank> <world> <port> <n> [timeout_s] [root_ok]
// Rank 0 owns a recognisable 128-byte token; every rank prints "ok <hex of the token>" or "error <text>".
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rendezvous.hpp"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), port = std::atoi(argv[3]);
  const int32_t sig[3] = {std::atoi(argv[4]), 100, 32};
  const int timeout = argc > 5 ? std::atoi(argv[5]) : 20;
  const bool root_ok = argc > 6 ? std::atoi(argv[6]) != 0 : true;
  char token[nbx_rendezvous::kTokenBytes];
  for (size_t i = 0; i < sizeof token; ++i) token[i] = rank == 0 ? (char)(i * 7 + 3) : 0;
  std::string err;
  if (!nbx_rendezvous::exchange(rank, world, "127.0.0.1", port, sig, token, timeout, &err, root_ok)) {
    std::printf("error %s\n", err.c_str());
    return 1;
  }
  std::printf("ok ");
  for (size_t i = 0; i < sizeof token; ++i) std::printf("%02x", (unsigned char)token[i]);
  std::printf("\n");
  return 0;
}
