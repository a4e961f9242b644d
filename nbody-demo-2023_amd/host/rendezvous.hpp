// rendezvous.hpp -- start-up exchange for the one-process-per-GPU mode of the drop-in.
//
// The reference's multi-process mode rests on MPI (MPI_Init + MPI_Bcast of npp_global, ver5_all/GSimulation.cpp:93-109);
// this image has none, and the only thing the ranks must share before RCCL can take over is the 128-byte communicator
// token (ncclUniqueId -> nbx_comm_unique_id / nbx_group_create_rank, include/nbx.h).  One TCP round does it:
//   every rank r > 0 connects to rank 0 (retrying while rank 0 is still starting), sends a fixed-size hello
//   {magic, rank, world, job signature}, and receives {status, token};
//   rank 0 accepts connections until every other rank of the job has said a well-formed hello (anything else on the port is
//   dropped; a hello must arrive within a second of its connection), checks them all (same world, distinct ranks in range, same
//   job signature: n, steps, precision) and only THEN answers -- every rank gets the same verdict, so a mis-launched or missing
//   rank refuses the whole job at once instead of leaving the ranks that were already accepted inside ncclCommInitRank.  A wrong or
//   duplicate hello does NOT take a seat: once the job is refused rank 0 keeps listening for a short grace period, so that the
//   ranks still on their way are told "refused" too instead of finding the port closed and retrying for their whole timeout.
// No data path runs over these sockets.  Environment (GSimulation::read_world_env): NBODY_WORLD / NBODY_RANK / NBODY_LOCAL_RANK,
// NBODY_MASTER_ADDR / NBODY_MASTER_PORT (default 127.0.0.1:29417); torchrun's WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR /
// MASTER_PORT are honoured only with NBODY_USE_TORCHRUN_ENV=1 (they are generic: a lone nbody.x that merely inherits them stays a
// one-process run).
#ifndef NBX_HOST_RENDEZVOUS_HPP
#define NBX_HOST_RENDEZVOUS_HPP

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace nbx_rendezvous {

const uint32_t kMagic = 0x4e425831u;  // "NBX1"
const size_t kTokenBytes = 128;       // NBX_UNIQUE_ID_BYTES

struct Hello {
  uint32_t magic;
  int32_t rank, world;
  int32_t sig[3];  // job signature: n, steps, precision
};
struct Reply {
  int32_t status;  // 0 ok, 1 refused
  char token[kTokenBytes];
};

inline bool send_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n > 0) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
inline bool recv_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n > 0) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
inline void set_timeouts(int fd, int seconds) {
  timeval tv; tv.tv_sec = seconds; tv.tv_usec = 0;
  ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}

// Rank 0 passes the token in `token` (kTokenBytes); every other rank receives it there.  Returns false with *err set
// on any failure, after at most ~timeout_s seconds: nobody blocks for ever on a rank that never came up.
// root_ok == false (rank 0 only): rank 0 could not produce a token; it still answers every rank -- with a refusal -- so that
// the job ends at once instead of every other rank waiting out its timeout.
inline bool exchange(int rank, int world, const std::string& addr, int port, const int32_t sig[3], char* token, int timeout_s,
                     std::string* err, bool root_ok = true) {
  using clock = std::chrono::steady_clock;
  const clock::time_point deadline = clock::now() + std::chrono::seconds(timeout_s);
  if (world <= 1) return true;
  if (rank == 0) {
    const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) { *err = "socket() failed"; return false; }
    int one = 1;
    ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in sa; std::memset(&sa, 0, sizeof sa);
    sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port); sa.sin_addr.s_addr = htonl(INADDR_ANY);
    if (::bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof sa) != 0 || ::listen(ls, world) != 0) {
      *err = "rank 0 cannot listen on port " + std::to_string(port) + ": " + std::strerror(errno);
      ::close(ls);
      return false;
    }
    // Two phases.  (1) Collect a hello from every other rank -- nobody is answered yet.  A connection that does not open
    // with the magic word (a port scanner, a health check, a stray client) is dropped and does not count.  (2) One verdict
    // for the whole job goes to every rank that said hello: a rank is never told "go" -- and left to wait inside
    // ncclCommInitRank for a communicator that cannot form -- when a later hello turns out to be wrong or missing.
    std::vector<char> seen((size_t)world, 0);
    seen[0] = 1;
    std::vector<int> fds;  // every connection that said a well-formed hello, good or bad: all get the verdict
    bool ok = true;
    int have = 1;          // distinct ranks of THIS job heard so far (a wrong or duplicate hello takes no seat)
    clock::time_point limit = deadline;  // shortened to a grace period once the job is refused
    const std::chrono::seconds grace(3);
    while (have < world) {
      const long left_ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(limit - clock::now()).count();
      if (left_ms <= 0) {
        if (ok) { *err = "rank 0 timed out waiting for " + std::to_string(world - have) + " rank(s)"; ok = false; }
        break;  // refused already: the grace period for ranks still on their way is over
      }
      timeval tv; tv.tv_sec = left_ms / 1000; tv.tv_usec = (left_ms % 1000) * 1000;
      ::setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);  // bounds accept()
      const int fd = ::accept(ls, NULL, NULL);
      if (fd < 0) {
        if (errno == EINTR) continue;
        if (ok) { *err = "rank 0 timed out waiting for " + std::to_string(world - have) + " rank(s)"; ok = false; }
        break;
      }
      set_timeouts(fd, 1);  // a rank sends its hello at once: a client that connects and says nothing costs one second, not ten
      Hello h;
      if (!recv_all(fd, &h, sizeof h) || h.magic != kMagic) { ::close(fd); continue; }  // not one of ours: ignore it
      set_timeouts(fd, 10);
      fds.push_back(fd);
      if (h.world != world || h.rank <= 0 || h.rank >= world || seen[(size_t)h.rank] || std::memcmp(h.sig, sig, sizeof h.sig) != 0) {
        if (ok) {
          *err = "rank " + std::to_string(h.rank) + " of " + std::to_string(h.world) + " does not belong to this job (n/steps/precision " +
                 std::to_string(h.sig[0]) + "/" + std::to_string(h.sig[1]) + "/" + std::to_string(h.sig[2]) + ", or a duplicate rank)";
          ok = false;
          // the job is refused as a whole; the ranks still on their way get the same answer if they arrive within the grace period
          if (clock::now() + grace < limit) limit = clock::now() + grace;
        }
        continue;
      }
      seen[(size_t)h.rank] = 1;
      have += 1;
    }
    ::close(ls);
    Reply r; std::memset(&r, 0, sizeof r);
    r.status = ok ? (root_ok ? 0 : 2) : 1;
    if (ok && root_ok) std::memcpy(r.token, token, kTokenBytes);
    for (size_t k = 0; k < fds.size(); ++k) {
      if (!send_all(fds[k], &r, sizeof r) && ok) { *err = "cannot answer a rank"; ok = false; }  // that rank times out by itself
      ::close(fds[k]);
    }
    if (ok && !root_ok) { *err = "rank 0 could not initialise; the other ranks were told to stop"; return false; }
    return ok;
  }
  // rank > 0: resolve once, then connect with retries until rank 0 listens
  addrinfo hints; std::memset(&hints, 0, sizeof hints);
  hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
  addrinfo* res = NULL;
  if (::getaddrinfo(addr.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res) {
    *err = "cannot resolve " + addr;
    return false;
  }
  int fd = -1;
  while (clock::now() < deadline) {
    fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
    if (fd >= 0) ::close(fd);
    fd = -1;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
  ::freeaddrinfo(res);
  if (fd < 0) { *err = "rank " + std::to_string(rank) + " could not reach rank 0 at " + addr + ":" + std::to_string(port); return false; }
  set_timeouts(fd, timeout_s);
  Hello h; h.magic = kMagic; h.rank = rank; h.world = world; std::memcpy(h.sig, sig, sizeof h.sig);
  Reply r;
  const bool ok = send_all(fd, &h, sizeof h) && recv_all(fd, &r, sizeof r);
  ::close(fd);
  if (!ok) { *err = "rank " + std::to_string(rank) + ": rendezvous with rank 0 broke off"; return false; }
  if (r.status == 2) { *err = "rank 0 could not initialise (see its message); stopping"; return false; }
  if (r.status != 0) { *err = "rank " + std::to_string(rank) + ": job refused by rank 0 (some rank has a different world size, n, steps or precision, is a duplicate, or never came)"; return false; }
  std::memcpy(token, r.token, kTokenBytes);
  return true;
}

}  // namespace nbx_rendezvous
#endif
