// Launch-plan replay: the per-step host loop of mi355/graph.py (Plan.run_forward / run_backward) in C.
//
// A plan is an array of pre-resolved launches: (launcher, its arguments as 64-bit slots, flags).  The Python side resolves
// every pointer ONCE per (plan, stream) and hands the table over; a training step is then two or three calls into this file
// instead of ~320 ctypes calls.  The launcher is reached through a typed trampoline generated per entry point of
// include/mi355conv.h (build/plan_table.inc lists them), so no signature is ever guessed: every slot is converted to the
// parameter type the prototype declares.
//
// Streams: launches flagged SIDE (weight gradients: they only feed the optimiser) carry the plan's side stream in their
// arguments; when a SIDE group starts, the side stream waits for an event recorded on the main stream at that point (fork),
// mi355_plan_join makes the main stream wait for the side stream.  Nothing here synchronises with the host.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include <exception>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mi355conv.h"

void mi355_set_error(const char* fmt, ...);

namespace {

template <typename T>
inline T unpack(uint64_t v) {
  if constexpr (std::is_pointer<T>::value) {
    return reinterpret_cast<T>(static_cast<uintptr_t>(v));
  } else if constexpr (std::is_same<T, float>::value) {
    const uint32_t b = static_cast<uint32_t>(v);
    float f;
    memcpy(&f, &b, 4);
    return f;
  } else if constexpr (std::is_same<T, double>::value) {
    double d;
    memcpy(&d, &v, 8);
    return d;
  } else {
    static_assert(std::is_integral<T>::value, "launcher parameter is neither pointer, float nor integer");
    return static_cast<T>(static_cast<int64_t>(v));
  }
}

template <typename... A, size_t... I>
inline int call_with(int (*fn)(A...), const uint64_t* a, std::index_sequence<I...>) {
  return fn(unpack<A>(a[I])...);
}
template <typename... A>
inline int call(int (*fn)(A...), const uint64_t* a) {
  return call_with(fn, a, std::index_sequence_for<A...>{});
}
template <typename... A>
constexpr int arity(int (*)(A...)) { return (int)sizeof...(A); }

typedef int (*tramp_t)(const uint64_t*);
struct Entry { const char* name; tramp_t tramp; int nargs; };

#define MI355_TRAMP(fn) {#fn, [](const uint64_t* a) -> int { return call(&fn, a); }, arity(&fn)},
const Entry kTable[] = {
#include "build/plan_table.inc"
};
#undef MI355_TRAMP

const Entry* find(const char* name) {
  for (const Entry& e : kTable)
    if (!strcmp(e.name, name)) return &e;
  return nullptr;
}

enum { FLAG_SIDE = 1 };

struct Launch {
  const Entry* e = nullptr;
  int flags = 0;
  uint32_t off = 0;        // first slot in Plan::args
};

struct Plan {
  std::vector<Launch> l;
  std::vector<uint64_t> args;
  std::vector<hipEvent_t> fork_ev;      // one per SIDE-group start, created on first use
  hipEvent_t join_ev = nullptr;
  int last_index = -1;
};

}  // namespace

extern "C" int mi355_plan_arity(const char* name) {
  const Entry* e = name ? find(name) : nullptr;
  return e ? e->nargs : -1;
}

extern "C" void* mi355_plan_create(int n) {
  if (n < 0) return nullptr;
  try {
    Plan* p = new Plan();
    p->l.resize(n);
    return p;
  } catch (const std::exception&) {                   // no C++ exception crosses the C ABI
    return nullptr;
  }
}

extern "C" int mi355_plan_set(void* plan, int i, const char* name, const uint64_t* args, int nargs, int flags) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p || i < 0 || i >= (int)p->l.size() || !name || (nargs > 0 && !args)) {
    mi355_set_error("plan_set: bad arguments");
    return MI355_ERR_ARG;
  }
  const Entry* e = find(name);
  if (!e) {
    mi355_set_error("plan_set: %s is not an entry point of mi355conv.h", name);
    return MI355_ERR_ARG;
  }
  if (e->nargs != nargs) {
    mi355_set_error("plan_set: %s takes %d arguments, got %d", name, e->nargs, nargs);
    return MI355_ERR_ARG;
  }
  try {
    const uint32_t off = (uint32_t)p->args.size();
    p->args.insert(p->args.end(), args, args + nargs);
    p->l[i].e = e;
    p->l[i].flags = flags;
    p->l[i].off = off;
  } catch (const std::exception& ex) {
    mi355_set_error("plan_set: %s", ex.what());
    return MI355_ERR_ARG;
  }
  return MI355_OK;
}

// Overwrite one argument slot (the network-input pointer of the first launch changes with every batch).
extern "C" int mi355_plan_patch(void* plan, int i, int arg, uint64_t value) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p || i < 0 || i >= (int)p->l.size() || !p->l[i].e || arg < 0 || arg >= p->l[i].e->nargs) {
    mi355_set_error("plan_patch: bad arguments");
    return MI355_ERR_ARG;
  }
  p->args[p->l[i].off + arg] = value;
  return MI355_OK;
}

extern "C" int mi355_plan_run(void* plan, int first, int last, mi355_stream_t main_stream, mi355_stream_t side_stream) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p || first < 0 || last > (int)p->l.size() || first > last) {
    mi355_set_error("plan_run: bad range [%d, %d)", first, last);
    return MI355_ERR_ARG;
  }
  bool prev_side = false;
  int fork = 0;
  for (int i = 0; i < first; ++i) {        // fork events are indexed by their position in the whole plan
    const bool side = (p->l[i].flags & FLAG_SIDE) != 0;
    if (side && !prev_side) ++fork;
    prev_side = side;
  }
  prev_side = false;                       // a range always re-establishes the dependency of its first SIDE group
  for (int i = first; i < last; ++i) {
    const Launch& l = p->l[i];
    if (!l.e) {
      mi355_set_error("plan_run: launch %d was never set", i);
      p->last_index = i;
      return MI355_ERR_ARG;
    }
    const bool side = side_stream && (l.flags & FLAG_SIDE);
    if (side && !prev_side) {              // fork: the side group may start once everything issued so far on main is done
      if ((int)p->fork_ev.size() <= fork) {
        try {
          p->fork_ev.resize(fork + 1, nullptr);
        } catch (const std::exception& ex) {
          mi355_set_error("plan_run: %s", ex.what());
          p->last_index = i;
          return MI355_ERR_ARG;
        }
      }
      hipEvent_t& ev = p->fork_ev[fork];
      hipError_t e = hipSuccess;
      if (!ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(ev, (hipStream_t)main_stream);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)side_stream, ev, 0);
      if (e != hipSuccess) {
        mi355_set_error("plan_run: stream fork failed: %s", hipGetErrorString(e));
        p->last_index = i;
        return (int)e;
      }
    }
    if ((l.flags & FLAG_SIDE) && !prev_side) ++fork;
    prev_side = (l.flags & FLAG_SIDE) != 0;
    const int rc = l.e->tramp(p->args.data() + l.off);
    if (rc) {
      p->last_index = i;
      return rc;
    }
  }
  return MI355_OK;
}

extern "C" int mi355_plan_join(void* plan, mi355_stream_t main_stream, mi355_stream_t side_stream) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p) return MI355_ERR_ARG;
  if (!side_stream) return MI355_OK;
  hipError_t e = hipSuccess;
  if (!p->join_ev) e = hipEventCreateWithFlags(&p->join_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(p->join_ev, (hipStream_t)side_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)main_stream, p->join_ev, 0);
  if (e != hipSuccess) {
    mi355_set_error("plan_join failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return MI355_OK;
}

extern "C" int mi355_plan_last_index(void* plan) { return plan ? static_cast<Plan*>(plan)->last_index : -1; }

extern "C" int mi355_plan_destroy(void* plan) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p) return MI355_OK;
  for (hipEvent_t ev : p->fork_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (p->join_ev) (void)hipEventDestroy(p->join_ev);
  delete p;
  return MI355_OK;
}
