// mcout.hh -- sample sink with the public interface of the reference's MCout (src/mcout.hh:32-50):
// rows of (np parameters, log-likelihood), appended per (step, chain), dumped as text by rank 0.
// MCPar::run fills it in bulk (add_rows) from the engine's HBM-resident sample store at the points
// where the reference dumps output; row order and text format follow src/mcout.cc:30-48,129-145.
#ifndef MCPAR_AMD_MCOUT_HH_
#define MCPAR_AMD_MCOUT_HH_

#include <cstddef>
#include <iostream>
#include <memory>
#include <new>
#include <vector>

#include "mpi_compat.hh"

MCPAR_ABI_NAMESPACE_BEGIN

// std::vector<float>::resize writes a zero into every new element: 4.5 GB of zeros for a 65 536-chain x 1000-sample run
// before a single row arrives, each of which add()/add_rows() overwrites.  With this allocator resize leaves them alone.
template <class T>
struct mcout_default_init : std::allocator<T> {
  template <class U> struct rebind { typedef mcout_default_init<U> other; };
  mcout_default_init() {}
  template <class U> mcout_default_init(const mcout_default_init<U> &) {}
  template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
  template <class U, class A> void construct(U *p, const A &a) { ::new (static_cast<void *>(p)) U(a); }
};

class MCout {
public:
  MCout(int np, std::ostream *aoutstream, MPI_Comm acomm);

  // ---- filling -------------------------------------------------------------------------------
  // reserve room for nsamp more rows (call before a loop that adds nsamp samples)
  void newsamps(int nsamp)
  {
    capacity_rows_ += nsamp;
    rows_.resize(rows_.size() + static_cast<std::size_t>(nsamp) * width_);
  }
  void add(const float *pv, float lval);                   // one row
  // nrows rows already in (np+1)-column layout; track_best = false: the caller reports the maximum itself (note_best)
  void add_rows(const float *rows, std::size_t nrows, bool track_best = true);

  // ---- inspection ----------------------------------------------------------------------------
  int size(void) const { return stored_rows_; }            // rows stored
  int maxsize(void) const { return capacity_rows_; }       // rows reserved
  int ncol(void) { return width_; }                        // np + 1
  int vsize(void) const { return static_cast<int>(rows_.size()); }
  const float *getpset(int i) const { return &rows_[static_cast<std::size_t>(i) * width_]; }
  float getlval(int i) const { return rows_[static_cast<std::size_t>(i + 1) * width_ - 1]; }

  // ---- output --------------------------------------------------------------------------------
  void output();                       // print the rows added since the last output()/collect()
  float *collect(std::size_t *ntot);   // rank 0: new[] buffer of all ranks' new rows (caller deletes)
  void rewind(void) { flushed_ = 0; }  // make every stored row "new" again
  // addition: output() writes the rows as raw little-endian float32 (np+1 per row, same order) instead of text --
  // the text formatting of src/mcout.cc:41-45 is 65-87 % of the reference's wall time
  void binary(bool on) { binary_ = on; }
  // addition: nothing is stored -- MCPar::run hands every block of rows to the stream as text formatted on the GPU
  // (mcx_set_text_sink: the same characters, at the speed of the copy).  size(), getpset(), collect() then see no rows;
  // maxlike() still answers (the engine keeps the running maximum).  With several ranks the ranks' texts of a block
  // are written in rank order, the row order of a dump.
  void text_only(bool on) { text_only_ = on; }
  bool text_only(void) const { return text_only_; }
  void write_text(const char *text, std::size_t nbytes);              // (MCPar::run's text sink; collective)
  // addition: every dump -- the text of text_only(), the rows output() prints (as "%g" text, or raw under binary()) --
  // goes to this FILE instead of rank 0's stream, and with
  // several ranks every rank writes its own share of a block itself, at the byte offset an MPI_Exscan of the shares'
  // sizes gives it -- the file is byte for byte what the funnel through rank 0 (the scheme the reference calls a
  // stop-gap, src/mcout.cc:22-35) would have written, without 8 ranks' text squeezing through one loop.  The file
  // must be on a file system every rank sees (one node: any).  COLLECTIVE; an empty or null path closes the file and
  // switches it off (like the reference's MCout, this class has no destructor: close the file when the run is over).
  // false: the file could not be opened on some rank (nothing changes then).
  bool text_file(const char *path);
  // (MCPar::run, collective) do all ranks have their block's text?  One rank without it sends every rank to the row path
  bool all_ranks_agree(bool mine);
  // (MCPar::run) would output() print the rows exactly as printf("%g") does -- a stream in its default state, no
  // binary mode, nothing waiting to be printed?  Then the text of a block may come from the GPU next to its rows ...
  bool prints_plain_text(void) const;
  // ... and the rows stored so far count as printed
  void mark_flushed(void) { flushed_ = fill_; }
  void note_best(float lval, const float *params);                    // (the engine's running maximum)
  // COLLECTIVE: every rank of the communicator must call it.  Best sample over all ranks.
  const std::vector<float> &maxlike(float *lmax);

private:
  const int nparam_, width_;      // parameters per row, columns per row
  std::vector<float, mcout_default_init<float> > rows_;  // row-major storage (rows not yet added: unspecified content)
  std::size_t fill_, flushed_;    // elements written / elements already handed to output()
  int stored_rows_, capacity_rows_;
  float best_l_;                  // running maximum of the log-likelihood column
  std::vector<float> best_p_;     // and the parameters it was seen at
  std::ostream *sink_;            // rank 0 only
  MPI_Comm comm_;
  int rank_, nranks_;
  bool binary_, text_only_;
  int text_fd_;                   // text_file(): this rank's descriptor of the shared file, or -1
  unsigned long long text_pos_;   // bytes of it written by all ranks so far
  void note_row(const float *row);
  void output_to_file(void);      // output() under text_file(): every rank's own rows, text or binary, into the file
};

MCPAR_ABI_NAMESPACE_END

#endif
