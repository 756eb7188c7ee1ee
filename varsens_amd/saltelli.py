"""Host-side mirror of the reference's three classes, executing on the GPU.

``Sample``, ``Objective`` and ``Varsens`` keep the constructor signatures,
attribute names, result shapes and error behaviour of varsens/saltelli.py
(:62, :287, :545 of the reference) so that ``import varsens_amd as varsens`` is
a drop-in.  What differs is where the work happens:

* ``Sample`` holds a *design* on the GPU (vs_design): generator, index range,
  scaling spec and the seed-1 permutation.  M_1, M_2, N_j, N_nj are never
  stored; the attributes of those names materialise rows on demand through
  vs_sample_flat.
* ``Objective`` holds the 2n(k+1) objective values in HBM in the reference's
  flat block order, written by the fused generate->evaluate kernel (device
  objectives), or uploaded (opaque callables, pre-computed values, files).
* ``Varsens.compute_varsens`` is two kernel passes (first-order streaming
  reduction, second-order Gram sums) plus small finishing kernels; with a
  communicator the accumulator record is summed over ranks in between.

The one-shot constructor ``Varsens(f, scaling, k, n, verbose=False)`` works as
the reference's docstring intends; at the reference's HEAD it raises TypeError
because ``verbose`` is passed positionally into ``discard``/``objective_vals``
(varsens/saltelli.py:555,567).
"""
import os

import numpy

from . import _native
from . import objectives as _objectives
from . import scale as _scale

_default_ctx = {}

# largest array an attribute such as Sample.N_j may materialise on the host
HOST_MATERIALISE_LIMIT = int(os.environ.get("VARSENS_HOST_LIMIT_BYTES", 8 << 30))
# rows handed to an opaque Python objective per chunk: row-by-row callables / numpy-vectorised ones
_HOST_CHUNK_ROWS = 1 << 16
_HOST_CHUNK_ROWS_VECTORIZED = 1 << 18


def default_context(device=None):
    """The process-wide GPU context (one GPU per process)."""
    if device is None:
        device = int(os.environ.get("VARSENS_DEVICE", os.environ.get("LOCAL_RANK", 0)))
    if device not in _default_ctx:
        _default_ctx[device] = _native.Context(device)
    return _default_ctx[device]


_perm_cache = {}


def seed1_permutation(n):
    """Row order the reference's ``numpy.random.seed(1); numpy.random.shuffle(M_2)``
    produces (varsens/saltelli.py:100-101): shuffled[r] = original[perm[r]].
    Computed by the engine's own restatement of numpy's legacy generator
    (vs_seed1_permutation: MT19937 seeded with 1, Fisher-Yates, masked rejection);
    numpy's global RNG is left alone (the reference reseeds it as a side effect).
    The walk is sequential (a few ns per row on the host), so the last few permutations
    are kept: repeated designs of one size (bootstraps) pay it once."""
    n = int(n)
    order = _perm_cache.get(n)
    if order is None:
        order = _native.seed1_permutation(n)
        order.setflags(write=False)
        if len(_perm_cache) >= 4:
            _perm_cache.pop(next(iter(_perm_cache)))
        _perm_cache[n] = order
    return order


def reseed_numpy_like_the_reference(n):
    """The reference's visible side effect (varsens/saltelli.py:100-101): ``numpy.random.seed(1)`` followed by
    ``numpy.random.shuffle(M_2)`` leaves numpy's GLOBAL generator in the state "seeded with 1, then n - 1 Fisher-Yates
    draws".  The engine takes its permutation from its own MT19937 (vs_seed1_permutation) and by default leaves numpy's
    generator alone; a caller whose later code draws from ``numpy.random`` and must see the reference's numbers asks for
    this (``Sample(..., reseed_numpy=True)`` or ``VARSENS_RESEED_NUMPY=1``).  The draws depend on the row count only --
    shuffling n bytes consumes what shuffling the (n, k) array does."""
    numpy.random.seed(1)
    numpy.random.shuffle(numpy.empty(int(n), dtype=numpy.uint8))


def sobol_directions(k):
    table = numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                                    "sobol_joe_kuo_1024x32.npy"))
    if k > table.shape[0]:
        raise Exception("the shipped Sobol table covers %d dimensions, k=%d asked" % (table.shape[0], k))
    return numpy.ascontiguousarray(table[:k])


def _shard_rows(n, shard):
    if shard is None:
        return 0, n
    begin, count = int(shard[0]), int(shard[1])
    if begin < 0 or count < 0 or begin + count > n:
        raise Exception("shard rows [%d,%d) are outside [0,%d)" % (begin, begin + count, n))
    return begin, count


class Sample(object):
    """The Saltelli design: M_1, M_2 and the column-swapped N_j, N_nj.

    Parameters are those of the reference (varsens/saltelli.py:62):
    ``Sample(k, n, scaling=None, discard=0, verbose=True, raw=None, **loadArgs)``
    plus keyword-only extras: ``generator`` ('halton' default, 'sobol'), ``reseed_numpy`` (default: the environment's
    ``VARSENS_RESEED_NUMPY`` = 1, else False: leave numpy's global generator as the reference leaves it, seeded with 1
    and advanced by the shuffle -- see ``reseed_numpy_like_the_reference``),
    ``context`` (a _native.Context), ``shard=(row_begin, row_count)`` (a rank of a multi-GPU job
    evaluates only these draws: only their entries of the seed-1 shuffle are uploaded, and the
    materialising attributes M_1 .. flat() are not available).

    ``scaling`` may be a ``scale.Linear/Power/Percentage/Magnitude/Identity``
    spec (applied on the GPU) or any callable on an (n,k) block (applied on the
    host to the two halves, as the reference does at :92,95).
    """

    def __init__(self, k, n, scaling=None, discard=0, verbose=True, raw=None, **loadArgs):
        generator = loadArgs.pop('generator', 'halton')
        reseed = loadArgs.pop('reseed_numpy', os.environ.get("VARSENS_RESEED_NUMPY", "0") == "1")
        self._ctx = loadArgs.pop('context', None) or default_context()
        # a rank of a multi-GPU job: only the shuffle entries of its own rows go to its GPU
        self._shard = loadArgs.pop('shard', None)
        self.k = int(k)
        self.n = int(n)
        self.scaling = scaling
        self.verbose = verbose
        self._host = {}       # name -> host array, for samples loaded from a flat file
        self._cache = {}      # attribute name -> materialised (read-only) host array; a design never changes
        self._design = None
        k, n = self.k, self.n

        unit = None  # host (2n,k) unit-cube points, when they come from outside
        if raw is not None:
            if self.verbose: print("Using provided raw sample")
            unit = numpy.asarray(raw)
            if unit.shape != (2 * n, k):
                raise Exception("Raw sample dimensions do not match specified dimensions")
        elif loadArgs:
            if self.verbose: print("Loading Sample from loadArgs")
            unit = self.load(**loadArgs)
            if unit.shape == (2 * n * (1 + k), k):
                return
        else:
            if self.verbose: print("Generating Low Discrepancy Sequence")
            if not self.scaling:
                raise Exception("Generating a fresh sample space requires that a 'scaling' function be defined.")
            if generator not in ('halton', 'sobol'):
                raise Exception("generator must be 'halton' or 'sobol'")

        if scaling is None:
            if self.verbose: print("Defaulting to identity scaling of parameter space")
            self.scaling = _scale.Identity()

        if self.verbose: print("Eliminating correlations")
        if reseed:
            reseed_numpy_like_the_reference(n)
        # the seed-1 shuffle of M_2 (saltelli.py:100-101): the engine supplies it itself and keeps the device copy per
        # (n, slice) in the context -- a study that calls Varsens(objective, scaling, k, n) once per model uploads the
        # 4 n bytes once (VS_DESIGN_SEED1_SHUFFLE)
        shuffle = _shard_rows(n, self._shard) if self._shard is not None else (0, n)

        if generator == 'halton':
            gen, first = _native.GEN_HALTON, 1 + 20 * k + int(discard)   # saltelli.py:83: skip 20k + discard
        else:
            gen, first = _native.GEN_SOBOL, 4096 + int(discard)          # quantlib/sobolGen.cpp:50 skipTo(4096)
        dirs = sobol_directions(k) if (unit is None and generator == 'sobol') else None

        if isinstance(self.scaling, _scale.DeviceScale):
            mode, lo, hi = self.scaling.device_spec(k)
            if unit is None:
                self._design = self._ctx.design_create(gen, k, n, first, mode, lo, hi, None, None, dirs,
                                                       seed1_shuffle=shuffle)
            else:
                self._design = self._ctx.design_create(_native.GEN_RAW, k, n, 0, mode, lo, hi, None, unit, None,
                                                       seed1_shuffle=shuffle)
        else:
            # opaque callable: its own Python code has to see the two (n,k) halves (saltelli.py:92,95)
            if unit is None:
                if gen == _native.GEN_HALTON:
                    buf = self._ctx.halton_fill(k, first, 2 * n)
                else:
                    buf = self._ctx.sobol_fill(k, dirs, first, 2 * n)
                unit = self._ctx.download(buf, (2 * n, k))
                buf.free()
            if self.verbose: print("Generating M_1")
            m1 = numpy.asarray(self.scaling(unit[0:n, ...]), dtype=numpy.float64)
            if self.verbose: print("Generating M_2")
            m2 = numpy.asarray(self.scaling(unit[n:2 * n, ...]), dtype=numpy.float64)
            if m1.shape != (n, k) or m2.shape != (n, k):
                raise Exception("scaling must map an (n,k) block to an (n,k) block")
            self._design = self._ctx.design_create(_native.GEN_RAW, k, n, 0, _native.SCALE_IDENTITY, None, None,
                                                   None, numpy.concatenate([m1, m2], axis=0), None,
                                                   seed1_shuffle=shuffle)
        if self.verbose: print("...Sample Created.")

    # -- materialisation on demand --------------------------------------------------------
    def _rows(self, begin, count):
        """Rows [begin, begin+count) of the flat sample as a host (count,k) array."""
        if count * self.k * 8 > HOST_MATERIALISE_LIMIT:
            raise Exception("refusing to materialise %d x %d sample rows on the host (limit %d bytes, "
                            "VARSENS_HOST_LIMIT_BYTES); evaluate on the device instead" %
                            (count, self.k, HOST_MATERIALISE_LIMIT))
        if 'flat' in self._host:
            return self._host['flat'][begin:begin + count]
        # produced piece by piece on the device and copied out through the library's pinned ring, the kernel of one
        # piece under the copy of the one before (vs_sample_rows): no allocation, launch, copy and free per request
        return self._ctx.sample_rows(self._design, begin, count, self.k)

    def _row_chunks(self, spans):
        """The rows of `spans` = [(begin, count), ...] one span at a time, span i+1 being produced and copied while
        the caller works on span i (vs_sample_rows_begin / _wait: two pinned slots).  Yields read-only views of the
        library's memory, each valid until the next one is asked for."""
        if 'flat' in self._host:
            for begin, count in spans:
                yield self._host['flat'][begin:begin + count]
            return
        for begin, count in spans:
            if count * self.k * 8 > HOST_MATERIALISE_LIMIT:
                raise Exception("refusing to materialise %d x %d sample rows on the host (VARSENS_HOST_LIMIT_BYTES)" %
                                (count, self.k))
        ctx, design, k = self._ctx, self._design, self.k
        if spans:
            ctx.sample_rows_begin(design, spans[0][0], spans[0][1], 0)
        for i, (begin, count) in enumerate(spans):
            if i + 1 < len(spans):
                ctx.sample_rows_begin(design, spans[i + 1][0], spans[i + 1][1], (i + 1) & 1)
            yield ctx.sample_rows_wait(i & 1, count, k)

    def _attribute(self, name, begin, count, shape):
        """One launch and one download per attribute, however often it is read: the reference's own idiom is
        ``sample.M_1[i]`` inside a loop (varsens/saltelli.py:335-355, tests/test_sample.py:44-54).  The array is
        handed out read-only -- the design lives on the GPU, writing into a host copy could not change it."""
        out = self._cache.get(name)
        if out is None:
            out = self._rows(begin, count).reshape(shape)
            out.setflags(write=False)
            self._cache[name] = out
        return out

    @property
    def M_1(self):
        return self._attribute('M_1', 0, self.n, (self.n, self.k))

    @property
    def M_2(self):
        return self._attribute('M_2', self.n, self.n, (self.n, self.k))

    @property
    def N_j(self):
        return self._attribute('N_j', 2 * self.n, self.k * self.n, (self.k, self.n, self.k))

    @property
    def N_nj(self):
        return self._attribute('N_nj', (2 + self.k) * self.n, self.k * self.n, (self.k, self.n, self.k))

    def generate_N_j(self, M_1, M_2):
        """k copies of M_2, copy i carrying column i of M_1 (varsens/saltelli.py:112-125).
        Host helper kept for API parity; the kernels never build these arrays."""
        M_1 = numpy.asarray(M_1)
        M_2 = numpy.asarray(M_2)
        out = numpy.repeat(M_2[numpy.newaxis, ...], self.k, axis=0)
        cols = numpy.arange(self.k)
        out[cols, :, cols] = M_1.T
        return out

    def flat(self):
        """(2n(k+1), k) array in the order [M_1; M_2; N_j[0..k); N_nj[0..k)] (varsens/saltelli.py:127-160)."""
        if self.verbose: print("Flattening sample space...")
        return self._rows(0, 2 * self.n * (self.k + 1))

    # -- text files (varsens/saltelli.py:173-250) -------------------------------------------
    def export(self, outdir=os.getcwd(), prefix="sample", postfix=".txt", blocksize=float("inf"), delimiter="\t"):
        total = 2 * self.n * (self.k + 1)
        blocksize = total if blocksize > total else int(blocksize)
        prefix = "_".join(str(prefix).split())
        if prefix[-1] == "_": prefix = prefix[:-1]
        prefix = os.path.join(outdir, prefix)
        nFiles = int(numpy.ceil(float(total) / blocksize))
        for b in range(nFiles):
            name = "%s%s" % (prefix, postfix) if nFiles == 1 else "%s_%d%s" % (prefix, b + 1, postfix)
            if self.verbose: print("Writing to %s ..." % name)
            rows = self._rows(b * blocksize, min(blocksize, total - b * blocksize))
            _native.text_write(name, rows, delimiter)       # numpy.savetxt's bytes (saltelli.py:187,192), threaded

    def load(self, indir='', loadFile=None, prefix=None, postfix='.txt', nFiles=None, offset=1, delimiter='\t'):
        files = _file_list("sample", indir, loadFile, prefix, postfix, nFiles, offset)
        parts = []
        for name in files:
            if not os.path.isfile(name):
                raise Exception("Cannot find input file " + name)
            if self.verbose: print("Reading " + name + " ...")
            parts.append(_native.text_read(name, delimiter))   # numpy.loadtxt's values (saltelli.py:217), threaded
        x = numpy.vstack(parts)
        n, k = self.n, self.k
        if x.shape == (2 * n, k):
            if not self.scaling:
                raise Exception("Loading a pre-generated, unscaled sample space requires that a 'scaling' "
                                "function be defined.")
        elif x.shape == (2 * n * (1 + k), k):
            if self.verbose: print("Flattened sample detected")
            # already scaled and shuffled: the design is M_1, M_2 as they stand
            self._host['flat'] = x
            self._design = self._ctx.design_create(_native.GEN_RAW, k, n, 0, _native.SCALE_IDENTITY, None, None,
                                                   None, x[0:2 * n])
        else:
            raise Exception("Loaded sample has shape " + str(x.shape) + ". Must have shape (%d,%d) or (%d,%d)." %
                            (2 * n, k, 2 * n * (1 + k), k))
        return x


def _file_list(what, indir, loadFile, prefix, postfix, nFiles, offset):
    if loadFile:
        return [os.path.join(indir, loadFile)]
    if not prefix:
        raise Exception("Either 'loadFile' or 'prefix' are required to load %s from file." %
                        ("a sample" if what == "sample" else "an objective"))
    if not nFiles:
        raise Exception("Loading %s files with 'prefix' requires defining 'nFiles'." % what)
    if prefix[-1] != "_": prefix += "_"
    return [os.path.join(indir, prefix) + str(i) + postfix for i in range(offset, offset + nFiles)]


class Objective(object):
    """The 2n(k+1) objective values, resident in HBM.

    ``Objective(k, n, sample=None, objective_func=None, objective_vals=[], verbose=True, **loadArgs)``
    as varsens/saltelli.py:287.  ``objective_func`` may be a device objective
    (``varsens_amd.objectives``) or any Python callable on one sample row.
    Keyword-only extras: ``shard=(row_begin, row_count)`` restricts a rank to a
    slice of the n draws (multi-GPU), ``context``, ``vectorized=True`` (a host
    callable that maps a (rows,k) block to (rows,) or (rows,l) values is called
    once per block of rows instead of once per row).
    """

    def __init__(self, k, n, sample=None, objective_func=None, objective_vals=[], verbose=True, **loadArgs):
        self._ctx = loadArgs.pop('context', None) or (sample._ctx if sample is not None else default_context())
        shard = loadArgs.pop('shard', None)
        lazy = loadArgs.pop('lazy', None)    # device objectives only: evaluate when the values are first needed (None: below)
        self._vectorized = loadArgs.pop('vectorized', False)  # host callables: f takes a (rows,k) block at once
        self.k = k
        self.n = n
        self.sample = sample
        self.objective_func = objective_func
        self.verbose = verbose
        self._row_begin, self._rows = _shard_rows(int(n), shard)
        self._f = None          # device buffer, planar: [l][2k+2][rows] (include/varsens_hip.h)
        self._l = None
        self._nan_trim = False  # only values that come through load() are trimmed (saltelli.py:474-495)
        self._nan_rows = 0      # rows of this shard load() found a NaN in (counted on the device)
        self._host_cache = None
        self._params = None
        self._order = None      # device objectives: the context's evaluation order at construction

        if self.verbose: print("Generating Objective Values.")
        if len(objective_vals) > 0:
            self.load(objective_vals)
        elif loadArgs:
            self.load(**loadArgs)
        else:
            if not self.sample:
                raise Exception("Generating a fresh objective requires that a 'sample' be defined.")
            elif not self.objective_func:
                raise Exception("Generating a fresh objective requires that an 'objective_func' be defined.")
            if isinstance(objective_func, _objectives.DeviceObjective):
                self._l = int(objective_func.outputs(int(self.k)))
                self._order = getattr(self._ctx, "eval_order", "fast")
                if lazy is None:
                    # A built-in objective needs no compilation: its values are produced when somebody first asks for
                    # them (fM_1 ..., flat(), export()), so that Varsens(objective) -- the reference's tutorial route
                    # Sample -> Objective -> Varsens -- can take a fused kernel and never write them.  The parameters
                    # are checked now; objectives compiled at run time are built and evaluated here, as before.
                    # What will be computed is fixed NOW: the parameter table is copied (_device_objective) and the
                    # evaluation order of the context is recorded (_order) -- values asked for after the context was
                    # switched to the other order are still produced in this one.  (A context in strict order
                    # evaluates now, as the reference does.)
                    lazy = (objective_func.kind != _native.OBJ_SOURCE and
                            not isinstance(objective_func, _objectives.SeparableExpression) and
                            getattr(self._ctx, "eval_order", "fast") == "fast")
                    if lazy:
                        self._device_objective().free()
                if not lazy:
                    self._evaluate_on_device()
            else:
                self._evaluate_on_host()

    # -- evaluation -----------------------------------------------------------------------
    def _device_objective(self):
        k = int(self.k)
        spec = self.objective_func
        if isinstance(spec, _objectives.SeparableExpression):
            return self._ctx.objective_create_separable(k, spec.term_body, spec.fold_sum, spec.params(k))
        if spec.kind == _native.OBJ_SOURCE:
            return self._ctx.objective_create_source(k, self._l, spec.body, spec.params(k))
        if self._params is None:     # the parameter table as it was when the Objective was made (values may come later)
            self._params = numpy.array(spec.params(k), dtype=numpy.float64, copy=True)
        return self._ctx.objective_create(spec.kind, k, self._l, self._params)

    def _values(self):
        """The device buffer of f-values, evaluating now if that was deferred."""
        if self._f is None and isinstance(self.objective_func, _objectives.DeviceObjective):
            self._evaluate_on_device()
        return self._f

    def _evaluate_on_device(self):
        k = int(self.k)
        spec = self.objective_func
        self._l = int(spec.outputs(k))
        handle = self._device_objective()
        self._f = self._ctx.alloc(8 * 2 * (k + 1) * self._rows * self._l)
        with self._ctx.eval_order_scope(self._order):      # the order of the day the Objective was made
            self._ctx.eval(self.sample._design, handle, self._row_begin, self._rows, self._f)
            self._ctx.sync()
        handle.free()

    def _evaluate_on_host(self):
        """An opaque Python objective: the sample rows come from the GPU chunk by chunk -- chunk i+1 is generated and
        copied into pinned memory while the model runs on chunk i --, the calls happen here one row at a time as in
        varsens/saltelli.py:329-355 (or once per chunk with ``vectorized=True``), and the values go back to HBM in the
        reference's own order: the device transposes them into its layout (vs_upload_values)."""
        k, n = int(self.k), int(self.n)
        func = self.objective_func
        first = self.sample._rows(self._row_begin, 1)
        if self._vectorized:
            test = numpy.asarray(func(first))
            l = 1 if test.ndim <= 1 else test.shape[1]
        else:
            test = func(first[0])
            try:
                l = len(test)
            except TypeError:
                l = 1
        self._l = l
        rows = self._rows
        vals = numpy.empty((2 * k + 2, rows, l))
        total, step = (2 * k + 2) * rows, 0
        report = max(1, int(0.01 * total))
        chunk = _HOST_CHUNK_ROWS_VECTORIZED if self._vectorized else _HOST_CHUNK_ROWS
        # (each chunk lives in a pinned slot of the library: at most 64 MiB of rows, whatever k is)
        chunk = max(64, min(chunk, (64 << 20) // (8 * k)))
        spans, where = [], []
        for block in range(2 * k + 2):
            for c0 in range(0, rows, chunk):
                c1 = min(rows, c0 + chunk)
                spans.append((block * n + self._row_begin + c0, c1 - c0))
                where.append((block, c0, c1))
        for (block, c0, c1), pts in zip(where, self.sample._row_chunks(spans)):
            if self._vectorized:   # extension: one call per block of rows (numpy-vectorised models); `pts` is a
                # read-only view that lives for the duration of the call
                vals[block, c0:c1] = numpy.asarray(func(pts), dtype=numpy.float64).reshape(c1 - c0, l)
                continue
            pts = numpy.array(pts)    # the rows are the model's to keep, as the reference's sample.M_1[i] are
            for i in range(c1 - c0):
                vals[block, c0 + i] = func(pts[i])
                step += 1
                if self.verbose and step % report == 0: print(str(int(round(100. * step / total))) + "%")
        self._f = self._ctx.alloc(8 * 2 * (k + 1) * rows * l)
        self._ctx.upload_values(vals.reshape(-1, l), rows, k, 0, rows, self._f)

    # -- access ---------------------------------------------------------------------------
    def _blocks(self):
        """Host copy [2k+2][rows][l] with NaN rows removed when the values came through load()."""
        if self._host_cache is None:
            k = int(self.k)
            if 8 * (2 * k + 2) * self._rows * self._l > HOST_MATERIALISE_LIMIT:
                raise Exception("refusing to copy the objective values to the host (VARSENS_HOST_LIMIT_BYTES)")
            # the device turns its planes back into the reference's order on the way out (vs_download_values)
            v = self._ctx.download_values(self._values(), self._rows, k, self._l).reshape(2 * k + 2, self._rows, self._l)
            if self._nan_trim and self._nan_rows:              # (the device counted them when the values came in)
                drop = numpy.isnan(v[:, :, 0]).any(axis=0)     # first output only (saltelli.py:482)
                v = v[:, ~drop]
            self._host_cache = v
        return self._host_cache

    @property
    def fM_1(self):
        return self._blocks()[0]

    @property
    def fM_2(self):
        return self._blocks()[1]

    @property
    def fN_j(self):
        return self._blocks()[2:2 + int(self.k)]

    @property
    def fN_nj(self):
        return self._blocks()[2 + int(self.k):]

    def flat(self):
        """(rows*2(k+1), l) values in block order (varsens/saltelli.py:357-391)."""
        if self.verbose: print("Flattening objectives...")
        v = self._blocks()
        return v.reshape(-1, v.shape[2])

    # -- text files (varsens/saltelli.py:393-472) --------------------------------------------
    def export(self, outdir=os.getcwd(), prefix="objective", postfix=".txt", blocksize=float("inf")):
        f = self.flat()
        blocksize = len(f) if blocksize > len(f) else int(blocksize)
        prefix = "_".join(str(prefix).split())
        if prefix[-1] == "_": prefix = prefix[:-1]
        prefix = os.path.join(outdir, prefix)
        nFiles = int(numpy.ceil(float(len(f)) / blocksize))
        for b in range(nFiles):
            name = "%s%s" % (prefix, postfix) if nFiles == 1 else "%s_%d%s" % (prefix, b + 1, postfix)
            if self.verbose: print("Writing to %s ..." % name)
            _native.text_write(name, f[b * blocksize:(b + 1) * blocksize], " ")   # saltelli.py:407,412

    def load(self, obj_vals=[], indir='', loadFile=None, prefix=None, postfix='.txt', nFiles=None, offset=1,
             scaling=1.0):
        k, n = int(self.k), int(self.n)
        if len(obj_vals) > 0:
            x = numpy.asarray(obj_vals, dtype=numpy.float64)
        else:
            files = _file_list("objective", indir, loadFile, prefix, postfix, nFiles, offset)
            parts = []
            for name in files:
                if not os.path.isfile(name):
                    raise Exception("Cannot find input file " + name)
                if self.verbose: print("Reading " + name + " ...")
                part = _native.text_read(name)              # numpy.loadtxt(name) (saltelli.py:439) ...
                if 1 in part.shape:
                    part = part.reshape(-1)                 # ... which hands back 1-D for one column (or one line)
                parts.append(part)
            # one observable: a list of 1-D arrays; several: 2-D arrays (saltelli.py:443-446)
            x = numpy.hstack(parts) if parts[0].ndim == 1 else numpy.vstack(parts)
        if len(x) != 2 * n * (1 + k):
            raise Exception("Loaded objective has length " + str(len(x)) + ". Must have length %d." %
                            (2 * n * (1 + k)))
        if x.ndim == 1:
            x = x[:, numpy.newaxis]
        self._l = x.shape[1]
        # `x / scaling` (saltelli.py:451-468) happens on the device while the values are laid out: a scalar or one
        # divisor per output; anything else numpy can broadcast is divided here, as the reference does
        divide = None
        if not (numpy.isscalar(scaling) and scaling == 1.0):
            d = numpy.asarray(scaling, dtype=numpy.float64)
            if d.size == 1 or (d.size == self._l and d.shape in ((self._l,), (1, self._l))):
                divide = d.ravel()
            else:
                x = x / scaling
        # slicing into fM_1 .. fN_nj, the shard's rows, the transposition into the device layout and the NaN scan of
        # saltelli.py:474-486 are one call: the array goes through a pinned ring in pieces, a kernel lays each piece out
        self._f = self._ctx.alloc(8 * 2 * (k + 1) * self._rows * self._l)
        lost = self._ctx.upload_values(x, n, k, self._row_begin, self._rows, self._f, divide)
        self._nan_trim = True
        self._nan_rows = lost
        self._host_cache = None
        if lost > 0:
            print("WARNING: %d of %d objectives were NaN, %f%% loss" %
                  (lost, 2 * n * (1 + k), 100.0 * lost / (2 * n * (1 + k))))


def stream_rows_per_pass(request, max_rows, bytes_per_row, hbm_bytes):
    """Draws per streamed pass (0 = hold all f-values in HBM) for a job whose LARGEST shard has
    `max_rows` draws.  Pure host logic, identical on every rank by construction."""
    if request in (None, 0, False):
        return 0
    if request != 'auto':
        r = int(request)
        return r if 0 < r < max_rows else 0
    if bytes_per_row * max_rows <= 0.5 * hbm_bytes:
        return 0
    # passes of at most 8 GiB: large enough to fill the GPU many times over, small enough that the
    # allocator never leaves its fast path (the first 64-GiB hipMalloc of a process takes seconds)
    r = int(min(0.25 * hbm_bytes, 8 << 30) // bytes_per_row)
    return max(64, r - r % 64)


def stream_passes(max_rows, rows_per_pass):
    """Passes every rank walks (ranks with fewer rows leave their last slots empty)."""
    return max(1, -(-int(max_rows) // int(rows_per_pass)))


class Varsens(object):
    """Variance-based sensitivity by the method of Saltelli (2002).

    ``Varsens(objective, scaling_func=None, k=None, n=None, sample=None, verbose=True)``
    (varsens/saltelli.py:545).  Results: ``E_2``, ``var_y`` of shape (l,); ``U_j``,
    ``U_nj``, ``sens``, ``sens_t`` (k,l); ``sens_2``, ``sens_2n`` (k,l,k,l).
    Keyword-only extras: ``second_order`` (default True), ``comm`` (an object with
    ``world``, ``rank`` and ``allreduce(buffer, count)`` -- see varsens_amd.dist),
    ``context``.

    The contract of ``comm.allreduce(buffer, count)``: sum the FIRST ``count`` doubles of the device buffer over the
    ranks, in place, on the context's current stream.  ``count`` is smaller than the buffer: an analysis keeps
    ``[Gram sums | accumulator record | finished results]`` in one allocation, the exchange covers the first two
    windows (``_reduce_doubles``) and the finishing kernels write the third afterwards -- a communicator must not touch
    what lies behind ``count`` (tests/test_gpu_api.py::test_the_exchange_covers_a_prefix_of_the_result_buffer).

        >>> import varsens_amd as varsens                                  # doctest: +SKIP
        >>> v = varsens.Varsens(varsens.objectives.GFunction([0, .5, 3, 9, 99, 99]),
        ...                     varsens.scale.Identity(), 6, 1024, verbose=False)  # doctest: +SKIP
    """

    def __init__(self, objective, scaling_func=None, k=None, n=None, sample=None, verbose=True, **extra):
        self.verbose = verbose
        self._second_order = extra.pop('second_order', True)
        self._comm = extra.pop('comm', None)
        self._fused = extra.pop('fused', 'auto')   # 'auto' | True | False: fused evaluate+reduce kernel
        vectorized = extra.pop('vectorized', False)  # host callable takes a (rows,k) block (extension)
        ctx = extra.pop('context', None)
        shard = extra.pop('shard', None)
        # device objectives: evaluate and reduce this many draws at a time instead of holding all
        # 2n(k+1) values in HBM ('auto': only when they would not fit comfortably)
        self._rows_per_pass = extra.pop('rows_per_pass', 'auto')
        if extra:
            raise TypeError("unexpected keyword arguments: %s" % sorted(extra))
        if isinstance(sample, Sample):
            self.sample = sample
            self.k = sample.k
            self.n = sample.n
        elif k is not None and n is not None and scaling_func is not None:
            self.k = k
            self.n = n
            kw = dict(context=ctx) if ctx is not None else {}
            if shard is not None and isinstance(objective, _objectives.DeviceObjective):
                # the rank's kernels need only its own entries of the shuffle; a host callable is fed through
                # vs_sample_flat, whose N_j / N_nj rows read M_2 rows anywhere in the shuffle: whole design
                kw['shard'] = shard
            self.sample = Sample(k, n, scaling_func, verbose=verbose, **kw)
        elif not isinstance(objective, Objective):
            raise ValueError("Must specify sample, (k,n,scaling_func), or Objective object")

        if isinstance(objective, Objective):
            self.objective = objective
            self.k = objective.k
            self.n = objective.n
        else:
            kw = dict(shard=shard) if shard is not None else {}
            if vectorized:
                kw['vectorized'] = True
            self._max_rows = _shard_rows(int(self.n), shard)[1]
            self._stream_rows = self._streaming_rows(objective, shard, ctx)
            if self._fused_candidate(objective) or self._stream_rows:
                kw['lazy'] = True   # the values are only produced if somebody asks for them
            elif self._fused is True:
                raise Exception("fused=True needs a device objective (with second_order: k * outputs <= 128)")
            self.objective = Objective(self.k, self.n, self.sample, objective, verbose=verbose, **kw)
        self._ctx = self.objective._ctx
        self.compute_varsens()

    _stream_rows = 0
    _max_rows = 0

    def _max_rows_over_ranks(self, rows):
        """The largest shard of the job: every streaming decision is taken from it, so that all ranks
        stream (or none does) with the same pass count -- the all-reduce sizes depend on both."""
        comm = self._comm
        if comm is None or comm.world == 1:
            return int(rows)
        if hasattr(comm, 'max'):
            return int(comm.max(rows))
        from .dist import shard_rows      # a communicator without max(): the standard row split
        return max(int(rows), max(shard_rows(int(self.n), comm.world, r)[1] for r in range(comm.world)))

    def _streaming_rows(self, objective, shard, ctx):
        """Draws per pass when the f-values are to be streamed through HBM in row ranges, else 0.
        Rank-invariant: computed from the largest shard over the ranks."""
        if not isinstance(objective, _objectives.DeviceObjective):
            return 0
        k = int(self.k)
        rows = self._max_rows_over_ranks(_shard_rows(int(self.n), shard)[1])
        self._max_rows = rows
        hbm = (ctx or self.sample._ctx).info()['hbm_bytes']
        per_row = 8 * 2 * (k + 1) * int(objective.outputs(k))
        stream = stream_rows_per_pass(self._rows_per_pass, rows, per_row, hbm)
        if not stream and per_row * rows > 0.9 * hbm and not self._fused_candidate(objective):
            # rows_per_pass = 0 / None, or a count >= the largest shard, mean "resident": say so before a
            # hipMalloc of 2n(k+1) values fails
            raise Exception("rows_per_pass=%r keeps all %d draws of the largest shard resident: %.1f GB of objective "
                            "values do not fit the %.1f GB of this GPU; pass rows_per_pass='auto' or a smaller count"
                            % (self._rows_per_pass, rows, per_row * rows / 1e9, hbm / 1e9))
        return stream

    def _fused_candidate(self, objective):
        """A device objective whose f-values may never be needed: the engine is asked later whether its fused
        evaluate+reduce kernels cover the (design, objective) pair -- vs_eval_reduce_supported for the first order
        alone, vs_eval_reduce_both_supported (k l <= 128 index rows) when the second-order indices are wanted too."""
        if self._fused is False or not isinstance(objective, _objectives.DeviceObjective):
            return False
        if not self._second_order:
            return True
        return int(self.k) * int(objective.outputs(int(self.k))) <= 128

    def compute_varsens(self):
        """E_2, var_y, U_j, U_nj, sens, sens_t, sens_2, sens_2n (varsens/saltelli.py:572-622)."""
        # whatever still has to be evaluated (fused kernels, streamed row ranges, deferred values) is evaluated in the
        # order the Objective was made in, not in the one the context may have been switched to since
        order = self.objective._order or getattr(self._ctx, "eval_order", "fast")
        with self._ctx.eval_order_scope(order):
            return self._compute_varsens()

    def _compute_varsens(self):
        if self.verbose: print("Final sensitivity calculation")
        ctx, obj = self._ctx, self.objective
        k, n, l = int(self.k), int(self.n), int(obj._l)
        world = self._comm.world if self._comm is not None else 1
        rank = self._comm.rank if self._comm is not None else 0
        flags = _native.REDUCE_NAN_TRIM if obj._nan_trim else 0

        # Every decision below is the same on all ranks of a job (it depends on the design, the objective and the
        # largest shard only): the all-reduce buffers must have one size everywhere, also on a rank without rows.
        stream = bool(self._stream_rows) and obj._f is None
        fused = False
        # a fused kernel needs no f-values at all: preferred over streaming, unless the caller asked for row ranges
        # of an explicit size
        if obj._f is None and self._fused_candidate(obj.objective_func) and not (stream and self._rows_per_pass != 'auto'):
            handle = obj._device_objective()
            design = obj.sample._design
            fused = (ctx.eval_reduce_both_supported(design, handle) if self._second_order
                     else ctx.eval_reduce_supported(design, handle))
            if not fused:
                handle.free()
                if self._fused is True:
                    raise Exception("fused=True: the fused kernels do not cover this design / objective "
                                    "(see vs_eval_reduce_first_order / vs_eval_reduce_first_second_order in "
                                    "include/varsens_hip.h)")
        if stream and not fused:
            return self._compute_streamed(self._stream_rows)

        nacc = _native.acc_len(k, l, world)
        acc, gram, whole = self._result_buffers(nacc, k * l)
        release = None
        if fused and obj._rows > 0:
            # evaluate and reduce in one kernel: the 2n(k+1) values never reach HBM
            if self._second_order:
                ctx.eval_reduce_first_second_order(design, handle, obj._row_begin, obj._rows, acc, gram, rank, world)
            else:
                ctx.eval_reduce_first_order(design, handle, obj._row_begin, obj._rows, acc, rank, world)
            release = handle        # freed behind the download of the results, which waits for the kernel
        else:
            if fused:
                handle.free()   # a rank without rows: the empty record of the reduction kernels
            if self._second_order:
                # both reductions read the same blocks: one pass where the shape allows
                ctx.reduce_first_second_order(obj._values(), obj._rows, k, l, acc, gram, flags, rank, world)
            else:
                ctx.reduce_first_order(obj._values(), obj._rows, k, l, acc, flags, rank, world)
        self._finish(acc, nacc, gram, world, whole)
        if release is not None:
            release.free()

    def _result_buffers(self, nacc, K):
        """ONE device buffer per analysis: [S S^T, T T^T sums (second order) | accumulator record | finished results].
        The first two windows are a rank's whole contribution -- one all-reduce over `_reduce_doubles(nacc, K)` doubles
        from the buffer's start --, the last two come back to the host in one download (`_finish`)."""
        k, l = int(self.k), int(self.objective._l)
        ng = _native.gram_len(K) if self._second_order else 0
        nacc_even = nacc + (nacc & 1)       # the finished results start on a 16-byte boundary
        nout = _native.first_order_len(k, l) + (2 * K * K if self._second_order else 0)
        whole = self._ctx.alloc(8 * (ng + nacc_even + nout))
        gram = _native.DeviceView(whole, 0, 8 * ng) if self._second_order else None
        acc = _native.DeviceView(whole, 8 * ng, 8 * nacc)
        return acc, gram, whole

    def _reduce_doubles(self, nacc, K):
        return nacc + (_native.gram_len(K) if self._second_order else 0)

    def _finish(self, acc, nacc, gram, slots, whole):
        """Exchange, finishing kernels and the result attributes; `slots` = variance slots in the record."""
        ctx, obj = self._ctx, self.objective
        k, n, l = int(self.k), int(self.n), int(obj._l)
        world = self._comm.world if self._comm is not None else 1
        K = k * l
        if world > 1:
            self._comm.allreduce(whole, self._reduce_doubles(nacc, K))      # ONE all-reduce: Gram sums + record
        # the finished results of both orders behind the record in the same buffer [first order | sens_2 | sens_2n]:
        # one allocation and ONE download per analysis (a call at k = 6, n = 1024 is 0.12 ms of wall time, most of it
        # such round trips)
        nf = _native.first_order_len(k, l)
        ng = _native.gram_len(K) if self._second_order else 0
        nacc_even = nacc + (nacc & 1)
        at = 8 * (ng + nacc_even)
        first = _native.DeviceView(whole, at, 8 * nf)
        ctx.finish_first_order(acc, n, k, l, first, slots)
        if self._second_order:
            s2 = _native.DeviceView(whole, at + 8 * nf, 8 * K * K)
            s2n = _native.DeviceView(whole, at + 8 * (nf + K * K), 8 * K * K)
            ctx.finish_second_order(gram, first, n, k, l, s2, s2n)
        tail = _native.DeviceView(whole, 8 * ng, whole.nbytes - 8 * ng)
        both = ctx.download(tail, (tail.nbytes // 8,))           # (waits for everything launched before it)
        self.accumulators = both[:nacc].copy()
        host = both[nacc_even:]
        r = host[:nf].reshape(2 + 4 * k, l)
        self.E_2 = r[0].copy()
        self.var_y = r[1].copy()
        self.U_j = r[2:2 + k].copy()
        self.U_nj = r[2 + k:2 + 2 * k].copy()
        self.sens = r[2 + 2 * k:2 + 3 * k].copy()
        self.sens_t = r[2 + 3 * k:2 + 4 * k].copy()
        if self._second_order:
            self.sens_2 = host[nf:nf + K * K].reshape(k, l, k, l).copy()
            self.sens_2n = host[nf + K * K:].reshape(k, l, k, l).copy()
        whole.free()

    def _compute_streamed(self, rows_per_pass):
        """The same estimators with the f-values streamed through one HBM buffer in row ranges:
        every range is a variance slot of the record (as a rank is), the records and Gram matrices
        add up on the device.  For designs whose 2n(k+1) values exceed the memory of the GPU."""
        ctx, obj = self._ctx, self.objective
        k, n, l = int(self.k), int(self.n), int(obj._l)
        world = self._comm.world if self._comm is not None else 1
        rank = self._comm.rank if self._comm is not None else 0
        R = int(rows_per_pass)
        passes = stream_passes(self._max_rows, R)   # from the largest shard: the same on every rank
        if passes * R < obj._rows:
            raise Exception("streamed passes (%d x %d rows) do not cover this rank's %d rows" % (passes, R, obj._rows))
        slots = passes * world
        nacc = _native.acc_len(k, l, slots)
        K = k * l
        # per-pass results and their running totals, each record + Gram sums in one buffer (one add per pass,
        # one all-reduce at the end)
        acc, gram, pass_buf = self._result_buffers(nacc, K)
        total, gram_total, total_buf = self._result_buffers(nacc, K)
        ctx.memset(total_buf, 0)
        f = ctx.alloc(8 * 2 * (k + 1) * max(1, min(R, obj._rows)) * l)
        handle = obj._device_objective()
        for c in range(passes):
            begin = c * R
            count = min(R, obj._rows - begin)
            if count <= 0:
                break
            ctx.eval(obj.sample._design, handle, obj._row_begin + begin, count, f)
            if self._second_order:
                ctx.reduce_first_second_order(f, count, k, l, acc, gram, 0, rank * passes + c, slots)
            else:
                ctx.reduce_first_order(f, count, k, l, acc, 0, rank * passes + c, slots)
            ctx.accumulate(total_buf, pass_buf, self._reduce_doubles(nacc, K))
        ctx.sync()
        handle.free()
        f.free()
        pass_buf.free()
        self._finish(total, nacc, gram_total, slots, total_buf)
