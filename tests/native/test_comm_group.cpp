// Host-only test of large-velocity-power-spectrum_amd/csrc/comm_group.h (no RCCL, no GPU): a recording stand-in for the RCCL
// table fails its n-th send / receive / group call, and the test checks that the group was closed on every path, that nothing
// was issued after the first failure, and that the first failure is what the caller gets.  Built and run by
// tests/test_host_logic.py::test_exchange_group_is_closed_on_every_error_path (g++).
#include <cstdio>
#include <string>
#include <vector>

#include "comm_group.h"

struct Elem { float re, im; };

struct FakeApi {
  typedef int result_t;
  static int success() { return 0; }
  int fail_send = -1, fail_recv = -1, fail_start = 0, fail_end = 0;   // index of the failing call (-1: none)
  int sends = 0, recvs = 0, open = 0, starts = 0, ends = 0;
  std::vector<std::string> log;
  int GroupStart() { ++starts; if (fail_start) return 7; ++open; return 0; }
  int GroupEnd() { ++ends; --open; return fail_end ? 9 : 0; }
  int Send(const Elem* p, size_t n, int peer) {
    if (sends == fail_send) { ++sends; return 3; }
    ++sends; log.push_back("S" + std::to_string(peer) + ":" + std::to_string(n) + "@" + std::to_string((long long)(p - base)));
    return 0;
  }
  int Recv(Elem* p, size_t n, int peer) {
    if (recvs == fail_recv) { ++recvs; return 5; }
    ++recvs; log.push_back("R" + std::to_string(peer) + ":" + std::to_string(n) + "@" + std::to_string((long long)(p - base)));
    return 0;
  }
  const Elem* base = nullptr;
};

static int failures = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

int main() {
  const int ncomp = 3, world = 4;
  const size_t blk = 10;
  std::vector<Elem> mem(2 * ncomp * world * blk);
  Elem* sendp[3];
  Elem* recvp[3];
  for (int k = 0; k < ncomp; ++k) {
    sendp[k] = mem.data() + (size_t)(2 * k) * world * blk;
    recvp[k] = mem.data() + (size_t)(2 * k + 1) * world * blk;
  }
  const char* what = nullptr;
  {   // clean run: every (component, peer) pair once, blocks at h * blk, counts in floats
    FakeApi a; a.base = mem.data();
    CHECK(vps_exchange_chunk_group(a, sendp, recvp, ncomp, world, blk, (size_t)2, &what) == 0 && what == nullptr);
    CHECK(a.sends == ncomp * world && a.recvs == ncomp * world && a.starts == 1 && a.ends == 1 && a.open == 0);
    CHECK(a.log[0] == "S0:20@0" && a.log[1] == "R0:20@40" && a.log[2] == "S1:20@10" && a.log[3] == "R1:20@50");
    CHECK(a.log.back() == "R3:20@" + std::to_string((long long)((2 * 2 + 1) * world * blk + 3 * blk)));
  }
  for (int n = 0; n < ncomp * world; ++n) {   // the n-th send fails: group closed, nothing issued afterwards
    FakeApi a; a.base = mem.data(); a.fail_send = n;
    CHECK(vps_exchange_chunk_group(a, sendp, recvp, ncomp, world, blk, (size_t)2, &what) == 3);
    CHECK(what && std::string(what) == "ncclSend");
    CHECK(a.open == 0 && a.ends == 1 && a.sends == n + 1 && a.recvs == n && (int)a.log.size() == 2 * n);
  }
  for (int n = 0; n < ncomp * world; ++n) {   // the n-th receive fails
    FakeApi a; a.base = mem.data(); a.fail_recv = n;
    CHECK(vps_exchange_chunk_group(a, sendp, recvp, ncomp, world, blk, (size_t)2, &what) == 5);
    CHECK(what && std::string(what) == "ncclRecv");
    CHECK(a.open == 0 && a.ends == 1 && a.sends == n + 1 && a.recvs == n + 1 && (int)a.log.size() == 2 * n + 1);
  }
  {   // the group cannot be opened: nothing issued, nothing to close
    FakeApi a; a.base = mem.data(); a.fail_start = 1;
    CHECK(vps_exchange_chunk_group(a, sendp, recvp, ncomp, world, blk, (size_t)2, &what) == 7);
    CHECK(std::string(what) == "ncclGroupStart" && a.sends == 0 && a.recvs == 0 && a.ends == 0 && a.open == 0);
  }
  {   // the close itself fails: reported, and a send failure takes precedence over it
    FakeApi a; a.base = mem.data(); a.fail_end = 1;
    CHECK(vps_exchange_chunk_group(a, sendp, recvp, ncomp, world, blk, (size_t)2, &what) == 9 && std::string(what) == "ncclGroupEnd");
    FakeApi b; b.base = mem.data(); b.fail_end = 1; b.fail_send = 2;
    CHECK(vps_exchange_chunk_group(b, sendp, recvp, ncomp, world, blk, (size_t)2, &what) == 3 && std::string(what) == "ncclSend");
    CHECK(b.ends == 1);
  }
  if (failures) return 1;
  std::printf("ok\n");
  return 0;
}
