// mcx_plan.hip -- the schedule of one run: MCPar::run's two loops (src/mcpar.cc:55-97, 99-210) cut into the items
// mcx_run executes.  Pure host logic, no device: tests/test_plan_schedule.py checks it on the CPU.
#include "mcx_engine_internal.hpp"

// ---------------------------------------------------------------------------------------------
// Schedule of one run(): pure host logic (no device), exported as mcx_plan() so that it can be
// tested without a GPU.  It is what mcx_run executes.
// ---------------------------------------------------------------------------------------------

// one Philox draw per step for the whole job (the reference draws per rank: src/mcpar.cc:142-146)
static inline bool coin_is_remote(const PlanCfg &c, int isamp)
{
  if (isamp < c.sync) return false;  // :143-144
  if (c.pl >= 1.0f) return false;    // (rndlocal < 1 always: no coin to look at -- 5 us of host time per 1000 steps of a 0.4 ms job)
  const uint32_t t = c.tbase + (uint32_t)c.nburn + (uint32_t)isamp;
  const float rndlocal = u24(philox4x32_10(t, 0u, 0u, 0u, c.seed, ST_COIN).x);
  return !(rndlocal <= c.pl);  // :152
}

std::vector<mcx_plan_item> build_plan(const PlanCfg &c)
{
  std::vector<mcx_plan_item> p;
  p.reserve(64);
  auto add = [&](int kind, int first, int nsteps, int aux) { p.push_back(mcx_plan_item{kind, first, nsteps, aux}); };
  // burn-in (src/mcpar.cc:55-97): the tuner looks at the counters when isamp > irate, irate = 50, 100, ...
  int irate = 50;
  for (int isamp = 0; isamp < c.nburn;) {
    int last = irate + 1 < c.nburn ? irate + 1 : c.nburn - 1;
    if (last - isamp + 1 > c.maxseg) last = isamp + c.maxseg - 1;
    const int steps = last - isamp + 1, check = last > irate ? 1 : 0;
    add(MCX_PLAN_BURN_SEGMENT, isamp, steps, 0);
    add(MCX_PLAN_TUNER, last, steps, check);
    if (check) irate += 50;
    isamp = last + 1;
  }
  if (c.nsamp > 0) add(MCX_PLAN_INIT_MOMENTS, 0, 0, 0);  // :99-104
  const int outstep = c.nsamp > 50 ? c.nsamp / 10 : 5;  // :110
  // Exchange schedule.  The reference gathers at every sync point (isamp % SYNCSTEP == 0, :127-140),
  // but the gathered slots are read only by genRemote, and every gather overwrites all of them: a
  // gather that is followed by another gather before the next Murray step is dead.  Default (lazy):
  // snapshot this shard's slot at every sync point, gather the latest snapshot right before a Murray
  // step reads it (and once at the end) -- bit-identical results, fused segments may span sync points.
  // eager = the reference's schedule (each gather overlapped with the next segment).
  bool need_gather = false;
  for (int isamp = 0; isamp < c.nsamp;) {
    if (c.sink_block > 0 && isamp % c.sink_block == 0 && isamp > 0) add(MCX_PLAN_SINK, isamp, c.sink_block, 0);
    if (isamp % outstep == 0 && isamp > 0 && c.output_hook) add(MCX_PLAN_OUTPUT, isamp, 0, 0);  // :115-119
    if (c.sharded && isamp % c.sync == 0) {  // :127-140
      add(MCX_PLAN_PUBLISH, isamp, 0, 0);
      if (c.eager) add(MCX_PLAN_GATHER_BEGIN, isamp, 0, 0);
      else need_gather = true;
    }
    if (coin_is_remote(c, isamp)) {  // :152-159
      if (need_gather) {  // the slot holds the snapshot of the last sync point
        add(MCX_PLAN_GATHER_BEGIN, isamp, 0, 0);
        need_gather = false;
      }
      if (c.sharded) add(MCX_PLAN_GATHER_WAIT, isamp, 0, 0);
      add(MCX_PLAN_PUBLISH, isamp, 0, 0);  // own slot is always current (:205-208)
      add(MCX_PLAN_REMOTE_STEP, isamp, 1, 0);
      ++isamp;
      continue;
    }
    // run of local steps up to the next output dump / Murray step (/ sync point when eager or unfused)
    const bool span_sync = c.fused && c.sharded && !c.eager;
    int steps = 1;
    while (isamp + steps < c.nsamp && steps < c.maxseg) {
      const int nx = isamp + steps;
      if (nx % outstep == 0 && c.output_hook) break;
      if (c.sink_block > 0 && nx % c.sink_block == 0) break;
      if (c.sharded && !span_sync && nx % c.sync == 0) break;
      if (coin_is_remote(c, nx)) break;
      ++steps;
    }
    int snap_after = -1;
    if (span_sync) {  // last sync point strictly inside the segment: the kernel snapshots the slot there
      const int last = ((isamp + steps - 1) / c.sync) * c.sync;
      if (last > isamp) {
        snap_after = last - 1 - isamp;
        need_gather = true;
      }
    }
    add(MCX_PLAN_MAIN_SEGMENT, isamp, steps, snap_after);
    isamp += steps;
  }
  if (c.sink_block > 0 && c.nsamp > 0) add(MCX_PLAN_SINK, c.nsamp, c.nsamp - ((c.nsamp - 1) / c.sink_block) * c.sink_block, 0);
  if (need_gather) add(MCX_PLAN_GATHER_BEGIN, c.nsamp, 0, 0);  // remote slots end as of the last sync point
  if (c.sharded) add(MCX_PLAN_GATHER_WAIT, c.nsamp, 0, 0);
  add(MCX_PLAN_PUBLISH, c.nsamp, 0, 0);
  return p;
}

extern "C" int mcx_plan(int nsamp, int nburn, int sync, float pl, uint32_t seed, uint32_t tbase, int nshards,
                        int eager, int fused, int max_segment, int has_output_hook, int sink_block_steps,
                        mcx_plan_item *items, int max_items, int *nitems)
{
  if (nsamp < 0 || nburn < 0 || sync < 1 || nshards < 1 || max_segment < 1 || sink_block_steps < 0 || !nitems)
    return fail(MCX_ERR_INVALID, "bad arguments");
  const PlanCfg c = {nsamp, nburn, sync, pl, seed, tbase, nshards > 1, eager != 0, fused != 0, has_output_hook != 0, max_segment,
                     sink_block_steps};
  const std::vector<mcx_plan_item> p = build_plan(c);
  *nitems = (int)p.size();
  if (items)
    for (int i = 0; i < (int)p.size() && i < max_items; ++i) items[i] = p[(size_t)i];
  return MCX_OK;
}

