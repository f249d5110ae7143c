// One chunk of the slab exchange as ONE RCCL group, closed on every path.
//
// Replaces what scripts/parallel_optimized.py:365-368 does with four blocking comm.allgather calls: every rank sends block h
// of every component's send buffer to rank h and receives rank h's block into slot h of the receive buffer.
//
// A failing ncclSend / ncclRecv must not leave the group open: an open group swallows every later RCCL call of the process
// (they are queued, never issued), so the peers already inside the matching exchange would wait forever.  The group is
// therefore ALWAYS closed with GroupEnd and the FIRST failure is what the caller sees.  Templated on the API table so that
// tests/native/test_comm_group.cpp can drive it with a recording stand-in on a host without RCCL or a GPU.
#pragma once
#include <cstddef>

// Api: result_t, success(), GroupStart(), GroupEnd(), Send(ptr, count_floats, peer), Recv(ptr, count_floats, peer)
template <class Api, class Elem>
typename Api::result_t vps_exchange_chunk_group(Api& api, Elem* const* sendp, Elem* const* recvp, int ncomp, int world,
                                                size_t block_elems, size_t floats_per_elem, const char** failed_call) {
  typedef typename Api::result_t R;
  *failed_call = nullptr;
  R r = api.GroupStart();
  if (r != api.success()) {
    *failed_call = "ncclGroupStart";
    return r;               // nothing was opened
  }
  R first = api.success();
  for (int k = 0; k < ncomp && first == api.success(); ++k)
    for (int h = 0; h < world && first == api.success(); ++h) {
      const size_t n = block_elems * floats_per_elem;
      r = api.Send(sendp[k] + (size_t)h * block_elems, n, h);
      if (r != api.success()) {
        first = r;
        *failed_call = "ncclSend";
        break;
      }
      r = api.Recv(recvp[k] + (size_t)h * block_elems, n, h);
      if (r != api.success()) {
        first = r;
        *failed_call = "ncclRecv";
      }
    }
  const R e = api.GroupEnd();   // on every path
  if (first != api.success()) return first;
  if (e != api.success()) *failed_call = "ncclGroupEnd";
  return e;
}
