// mcx_sink.hip -- the role of MCout as the reference fills and dumps it (src/mcpar.cc:110-119, 176-182; src/mcout.cc):
// the streaming sample sink (a ring of blocks in HBM, staged out on a copy stream), its text side (the rows as the
// characters MCout::output prints, formatted on the device: mcx_text.hpp) and the text of rows kept in HBM.
#include "mcx_engine_internal.hpp"
#include "mcx_text.hpp"

// ---------------------------------------------------------------------------------------------
// Streaming sample sink (mcx_set_sink).  Block number seq (main-loop steps [done - nsteps, done)) has just been
// queued on the step stream: note its maximum-likelihood sample, then stage it out on the copy stream -- rows
// interleaved into MCout's layout on the device, one D2H into pinned memory -- while the step stream runs on.
// Two staging buffers: before block seq may take buffer seq % 2, the consumer is given block seq - 2 (the host
// waits for THAT copy only; the device ring holds SINK_RING = 4 blocks, so steps are never held up by a slot
// that is still being read as long as the consumer keeps up).
// ---------------------------------------------------------------------------------------------
// Text of block seq, in two moves.  (1) Its byte count has arrived (ev_copy), so the fields are formatted once more into
// place (mcx_text.hpp) on the copy stream -- one block after its rows were staged, while the block before it is still on
// its way to the host.  The staging buffer still holds the rows: the block that reuses it is queued two blocks later,
// behind this kernel on the same stream.  (2) The text leaves on a stream of its own as soon as the consumer has
// returned the one pinned buffer (pinning a second gigabyte costs a one-shot driver more than the overlap saves it):
// the copy of a block (PCIe, 5-6 ms for C3's 287 MB) runs under the formatting of the next.
static int sink_text_write(mcx_engine *e, int seq)
{
  const int b = seq & 1;
  HIPCHK(hipEventSynchronize(e->ev_copy[b]));
  const int first = seq * e->run_kb;
  const int kept = std::min(e->run_kb, (e->last_sink_total - first));
  const size_t total = (size_t)e->sink_text_total[b].p[0];
  const size_t count = (size_t)kept * e->nchain * (size_t)(e->nparam + 1), nwg = (count + BLOCK - 1) / BLOCK;
  e->sink_text_bytes[b] = total;
  e->sink_text_ok[b] = false;
  e->sink_text_written = seq + 1;
  const int arc = e->sink_text_dev[b].alloc(total + total / 8);
  if (arc != MCX_OK) return e->tfn ? arc : MCX_OK;  // (a row sink whose block's text found no memory gets its rows all the same)
  hipLaunchKernelGGL(k_text_write, dim3((unsigned)nwg), dim3(BLOCK), 0, e->cstream, (const float *)e->sink_stage[b].p,
                     (const float *)nullptr, count, e->nparam, (const unsigned long long *)e->sink_text_wg[b].p, e->sink_text_dev[b].p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev_write[b], e->cstream));
  e->sink_text_ok[b] = true;
  return MCX_OK;
}

static int sink_text_copy(mcx_engine *e, int seq)
{
  const int b = seq & 1;
  e->sink_text_copied = seq + 1;
  if (!e->sink_text_ok[b]) return MCX_OK;
  const size_t total = e->sink_text_bytes[b];
  if (total > e->sink_text_pin.n) {
    const int arc = e->sink_text_pin.alloc(total + total / 8);
    if (arc != MCX_OK) {
      e->sink_text_ok[b] = false;
      return e->tfn ? arc : MCX_OK;
    }
  }
  HIPCHK(hipStreamWaitEvent(e->tstream, e->ev_write[b], 0));
  HIPCHK(hipMemcpyAsync(e->sink_text_pin.p, e->sink_text_dev[b].p, total, hipMemcpyDeviceToHost, e->tstream));
  HIPCHK(hipEventRecord(e->ev_text, e->tstream));
  return MCX_OK;
}

static int sink_deliver(mcx_engine *e, int seq)
{
  MCXCHK(meet_release(e, false));  // (the consumer may take its time)
  const int b = seq & 1;
  HIPCHK(hipEventSynchronize(e->ev_copy[b]));
  const int first = seq * e->run_kb;  // kept steps before this block
  const int kept = std::min(e->run_kb, (e->last_sink_total - first));
  const bool texts = e->tfn || e->run_sink_text;
  if (texts) {
    if (seq >= e->sink_text_written) MCXCHK(sink_text_write(e, seq));  // (the run's last blocks: nothing came after them)
    if (seq >= e->sink_text_copied) MCXCHK(sink_text_copy(e, seq));
    if (e->sink_text_ok[b]) {
      HIPCHK(hipEventSynchronize(e->ev_text));
      if (e->tfn) {
        if (e->tfn(e->sctx, first, kept, e->sink_text_pin.p, e->sink_text_bytes[b]) != 0) return fail(MCX_ERR_INVALID, "sample sink failed");
      } else {
        e->cb_text = e->sink_text_pin.p;  // (row sink with MCX_OPT_SINK_TEXT: the callback asks mcx_sink_text for it)
        e->cb_text_bytes = e->sink_text_bytes[b];
      }
    }
    // (a row sink whose block's text found no memory gets its rows all the same: mcx_sink_text then says so)
  }
  int rc = 0;
  if (!e->tfn) {
    rc = e->sfn(e->sctx, first, kept, e->sink_pin[b].p);
    e->cb_text = nullptr;
    e->cb_text_bytes = 0;
  }
  // the pinned buffer is free again: the next block's text, formatted by now, may follow
  if (texts && rc == 0 && seq + 1 < e->sink_text_written && seq + 1 >= e->sink_text_copied) MCXCHK(sink_text_copy(e, seq + 1));
  if (rc != 0) return fail(MCX_ERR_INVALID, "sample sink failed");
  return MCX_OK;
}

int sink_block_done(mcx_engine *e, int done, int nsteps, int seq)
{
  const int n = e->nchain, d = e->nparam, b = seq & 1;
  const int first_step = done - nsteps;  // the block's first main-loop step: a multiple of the block length
  float *vx, *vl;
  samp_vbase(e, first_step, &vx, &vl);
  const size_t row0 = (size_t)(first_step / e->opt_stride);
  const size_t kept = (size_t)((done + e->opt_stride - 1) / e->opt_stride) - row0;
  const float *sx = vx + row0 * e->ntot, *sl = vl + row0 * n;
  const bool texts = e->tfn || e->run_sink_text;
  if (seq == 0) e->sink_text_written = e->sink_text_copied = 0;
  // running maximum (src/mcout.cc:140-144), on the step stream: cheap, and ordered before the slot's reuse
  hipLaunchKernelGGL(k_argmax_first, dim3(std::min<unsigned>(nblocks(kept * n), 1024u)), dim3(BLOCK), 0, e->stream, sl, kept * n, e->best_key.p);
  hipLaunchKernelGGL(k_best_update, dim3(1), dim3(BLOCK), 0, e->stream, e->best_key.p, sl, sx, d, e->best_row.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev_steps[b], e->stream));
  // the previous block's text: formatted and sent off now that its size is known (last_sink_total still counts up to it)
  if (texts && seq >= 1 && e->sink_text_written < seq) MCXCHK(sink_text_write(e, seq - 1));
  if (texts && seq == 1) MCXCHK(sink_text_copy(e, 0));  // (the first block's: nobody holds the pinned buffer yet)
  e->last_sink_total = (int)(row0 + kept);
  if (seq >= 2) MCXCHK(sink_deliver(e, seq - 2));  // frees staging buffer b
  HIPCHK(hipStreamWaitEvent(e->cstream, e->ev_steps[b], 0));
  hipLaunchKernelGGL(k_rows_interleave, dim3(nblocks(kept * n * (d + 1))), dim3(BLOCK), 0, e->cstream, sx, sl,
                     e->sink_stage[b].p, kept * n, d);
  HIPCHK(hipGetLastError());
  if (texts) {  // the size of the block's text (its two counting passes); sink_text_write / _copy place and send it
    const size_t count = kept * n * (size_t)(d + 1), nwg = (count + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_text_sizes, dim3((unsigned)nwg), dim3(BLOCK), 0, e->cstream, (const float *)e->sink_stage[b].p,
                       (const float *)nullptr, count, d, e->sink_text_wg[b].p);
    hipLaunchKernelGGL(k_text_scan, dim3(1), dim3(1024), 0, e->cstream, e->sink_text_wg[b].p, nwg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->sink_text_total[b].p, e->sink_text_wg[b].p + nwg, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->cstream));
  }
  if (!e->tfn)
    HIPCHK(hipMemcpyAsync(e->sink_pin[b].p, e->sink_stage[b].p, kept * n * (d + 1) * sizeof(float), hipMemcpyDeviceToHost, e->cstream));
  HIPCHK(hipEventRecord(e->ev_copy[b], e->cstream));
  // the ring: block seq + SINK_RING - 1 will overwrite the slot of block seq - 1, whose copy is already waited for
  // two blocks from now at the latest; with SINK_RING = 4 the host-side wait above is the only synchronisation
  return MCX_OK;
}

int sink_drain(mcx_engine *e, int nblocks_done)
{
  for (int seq = std::max(0, nblocks_done - 2); seq < nblocks_done; ++seq) MCXCHK(sink_deliver(e, seq));
  return MCX_OK;
}

extern "C" int mcx_set_sink(mcx_engine *e, mcx_sink_fn fn, void *ctx, int block_steps)
{
  if (!e || (fn && block_steps < 1)) return fail(MCX_ERR_INVALID, "bad arguments");
  if (e->pend.active) MCXCHK(enter(e));  // (an asynchronous run in flight is finished first)
  e->sfn = fn;
  e->tfn = nullptr;
  e->sctx = ctx;
  e->sink_block = fn ? block_steps : 0;
  return MCX_OK;
}

extern "C" int mcx_sink_text(mcx_engine *e, const char **text, size_t *nbytes)
{
  if (!e || !text || !nbytes) return fail(MCX_ERR_INVALID, "bad arguments");
  if (!e->cb_text) return fail(MCX_ERR_INVALID, "no block text: call it from a sink callback of a run with MCX_OPT_SINK_TEXT");
  *text = e->cb_text;
  *nbytes = e->cb_text_bytes;
  return MCX_OK;
}

extern "C" int mcx_set_text_sink(mcx_engine *e, mcx_text_sink_fn fn, void *ctx, int block_steps)
{
  if (!e || (fn && block_steps < 1)) return fail(MCX_ERR_INVALID, "bad arguments");
  if (e->pend.active) MCXCHK(enter(e));  // (an asynchronous run in flight is finished first)
  e->tfn = fn;
  e->sfn = nullptr;
  e->sctx = ctx;
  e->sink_block = fn ? block_steps : 0;
  return MCX_OK;
}

// rows on the device -> their text (mcx_text.hpp); sl == null: sx holds whole rows of d + 1 columns
static int text_of_rows(const float *sx, const float *sl, size_t count, int d, DevBuf<unsigned long long> &wg, DevBuf<char> &dev,
                        hipStream_t st, char *text, size_t capacity, size_t *nbytes)
{
  *nbytes = 0;
  if (count == 0) return MCX_OK;
  const size_t nwg = (count + BLOCK - 1) / BLOCK;
  if (nwg > 0x7fffffffu) return fail(MCX_ERR_INVALID, "too many rows for one call: ask for fewer at a time");
  MCXCHK(wg.alloc(nwg + 1));
  hipLaunchKernelGGL(k_text_sizes, dim3((unsigned)nwg), dim3(BLOCK), 0, st, sx, sl, count, d, wg.p);
  hipLaunchKernelGGL(k_text_scan, dim3(1), dim3(1024), 0, st, wg.p, nwg);
  HIPCHK(hipGetLastError());
  unsigned long long total = 0;
  HIPCHK(hipMemcpyAsync(&total, wg.p + nwg, sizeof total, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *nbytes = (size_t)total;
  if (!text) return MCX_OK;  // (the size only)
  if ((size_t)total > capacity) return fail(MCX_ERR_INVALID, "text buffer too small: %llu bytes needed, %zu given", total, capacity);
  MCXCHK(dev.alloc((size_t)total));
  hipLaunchKernelGGL(k_text_write, dim3((unsigned)nwg), dim3(BLOCK), 0, st, sx, sl, count, d, (const unsigned long long *)wg.p, dev.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(text, dev.p, (size_t)total, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return MCX_OK;
}

// The same rows as text: what MCout::output prints for them (src/mcout.cc:41-45), formatted on the device.
extern "C" int mcx_samples_text(mcx_engine *e, int first_step, int nsteps, char *text, size_t capacity, size_t *nbytes)
{
  MCXCHK(enter(e));
  if (!e || !nbytes || first_step < 0 || nsteps < 0 || (!text && capacity)) return fail(MCX_ERR_INVALID, "bad arguments");
  if (first_step + nsteps > e->samp_steps) return fail(MCX_ERR_INVALID, "steps [%d,%d) not in the sample store (%d steps)", first_step, first_step + nsteps, e->samp_steps);
  const size_t n = (size_t)e->nchain, d = (size_t)e->nparam;
  return text_of_rows(e->samp_x.p + (size_t)first_step * n * d, e->samp_ly.p + (size_t)first_step * n, (size_t)nsteps * n * (d + 1),
                      (int)d, e->text_wg, e->text_dev, e->stream, text, capacity, nbytes);
}

// any rows on the host (ncol columns each), e.g. what a sink received: uploaded, formatted, the text copied back
extern "C" int mcx_format_rows(const float *rows, size_t nrows, int ncol, char *text, size_t capacity, size_t *nbytes)
{
  if (!nbytes || ncol < 1 || (nrows && !rows) || (!text && capacity)) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  DevBuf<float> dr;
  DevBuf<unsigned long long> wg;
  DevBuf<char> dev;
  const size_t count = nrows * (size_t)ncol;
  int rc = dr.alloc(count);
  if (rc == MCX_OK && count && hipMemcpy(dr.p, rows, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(MCX_ERR_HIP, "hipMemcpy failed");
  if (rc == MCX_OK) rc = text_of_rows(dr.p, nullptr, count, ncol - 1, wg, dev, nullptr, text, capacity, nbytes);
  dr.release(); wg.release(); dev.release();
  return rc;
}

