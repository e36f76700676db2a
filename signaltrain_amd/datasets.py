"""Data sets of the training driver: SynthAudioDataSet (signaltrain/datasets.py:263-334, synthetic windows of an effect -- generated on the GPU),
AudioFileDataSet (:64-259, pre-recorded input / target pairs) and their device-resident loaders.  Items: (x f32[chunk], y[y_size], knobs f32[K])."""
import numpy as np
from torch.utils.data import Dataset

from . import audio


def do_augment(x, y, rand_invert=True):
    if rand_invert and np.random.choice([True, False]):
        x, y = -x, -y
    return x, y


def worker_init(worker_id):
    """datasets.py:54-61: reseed numpy per DataLoader worker (not reproducible by design)."""
    np.random.seed()


class SynthAudioDataSet(Dataset):
    """signaltrain/datasets.py:263-334: on-the-fly synthetic (x, y, knobs) items of an Effect.  Here the items are made ON THE GPU, a whole
    minibatch per call (batch_device; DeviceSynthLoader / DeviceRecycledDataSet iterate it for train.train); there is no per-item host
    generator behind __getitem__ -- the reference's CPU workers top out at ~100 windows/s per core, three to four orders of magnitude
    below the train step."""

    def __init__(self, chunk_size, effect, sr=44100, datapoints=8000, dtype=np.float32, recycle=False, y_size=None, augment=True):
        super().__init__()
        self.chunk_size, self.effect, self.sr, self.datapoints, self.dtype = chunk_size, effect, sr, datapoints, dtype
        self.recycle, self.num_knobs, self.augment = recycle, len(effect.knob_names), augment
        self.y_size = chunk_size if y_size is None else y_size
        # stream of the fused device feed: (seed, global window index).  The seed is drawn HERE from numpy's global generator, so it follows
        # np.random.seed(...) of the run (and the rank, train.seed_data_streams); the index counts the windows this dataset has produced
        self._feed_seed, self._feed_count = int(np.random.randint(0, 2 ** 31 - 1)), 0

    def __len__(self):
        return self.datapoints

    def __getitem__(self, idx):
        raise NotImplementedError("signaltrain_amd.SynthAudioDataSet generates minibatches on the GPU: use batch_device(), "
                                  "DeviceSynthLoader or DeviceRecycledDataSet (train.train does)")

    def batch_device(self, B, device="cuda:0", generator=None, chooser=-1):
        """Device-resident minibatch generated ON the GPU: input signals and knob settings by the batched
        device generators of audio_device.py (counter-free device RNG, no host loop), the sequential compressor as one HIP launch
        for the whole batch (st_compressor_4c) -- or, for the comp_4c effects on a ROCm device, everything in ONE launch (st_synth_comp4c).
        chooser: force one signal family (tests).  Returns (x, y, knobs) torch tensors."""
        import torch
        from . import audio_device
        device = torch.device(device)
        fused = generator is None and device.type == "cuda" and len(self.effect.knob_ranges) == 4 \
            and type(self.effect).__name__.startswith("Compressor_4c")
        if fused:
            # ONE launch per minibatch (st_synth_comp4c, csrc/st_feed.h): signals, knobs, compressor and the polarity flip of the pair.
            # Counter-based generator: the stream is (seed, global window index) -- the seed follows np.random.seed(...) of the run
            # (and so the rank, train.seed_data_streams), the index counts the windows this dataset has produced.
            import ctypes as C
            from . import _lib
            x = torch.empty(B, self.chunk_size, dtype=torch.float32, device=device)
            y = torch.empty(B, self.y_size, dtype=torch.float32, device=device)
            kn = torch.empty(B, 4, dtype=torch.float32, device=device)
            pink = None
            if self.chunk_size > 8192 or (self.chunk_size & (self.chunk_size - 1)):      # beyond the in-kernel FFT: rocFFT makes the 1/f noise
                g = getattr(self, "_dev_gen", None)
                if g is None or g.device != device:
                    g = torch.Generator(device=device); g.manual_seed(self._feed_seed); self._dev_gen = g
                pink = audio_device.pinknoise(B, self.chunk_size, g, device).contiguous()
            lo = (C.c_float * 4)(*[float(v) for v in self.effect.knob_ranges[:, 0]])
            hi = (C.c_float * 4)(*[float(v) for v in self.effect.knob_ranges[:, 1]])
            with torch.cuda.device(device):
                _lib.check(_lib.load().st_synth_comp4c(self._feed_seed, self._feed_count, B, self.chunk_size, self.y_size, 4, float(self.sr), lo, hi,
                                                       1 if self.augment else 0, int(chooser), _lib.ptr(pink), _lib.ptr(x), _lib.ptr(y), _lib.ptr(kn),
                                                       _lib.ptr(torch.empty(B * (self.chunk_size + 4), dtype=torch.float32, device=device)),
                                                       C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "st_synth_comp4c")
            self._feed_count += B
            return x, y, kn
        if True:
            gen = generator
            if gen is None:
                gen = getattr(self, "_dev_gen", None)
                if gen is None or gen.device != device:
                    gen = torch.Generator(device=device); gen.manual_seed(int(np.random.randint(0, 2 ** 31 - 1)))    # follows np.random.seed(...) of the run
                    self._dev_gen = gen
            x, _ = audio_device.synth_input_batch(B, self.chunk_size, self.sr, gen, device, chooser=(None if chooser < 0 else chooser))
            kn = audio_device.random_ends(B, len(self.effect.knob_ranges), gen, device) - 0.5
        y = self.effect.go_device(x, kn, self.y_size)
        if self.augment:                                   # do_augment: random polarity flip of the pair (datasets.py:27-29)
            sgn = torch.where(torch.rand(B, 1, device=device, generator=gen) < 0.5, -1.0, 1.0)
            x, y = x * sgn, y * sgn
        return x, y, kn


class DeviceSynthLoader:
    """The reference's TRAINING feed (SynthAudioDataSet without recycling behind a shuffling DataLoader, train.py:233-248:
    every item is generated on the fly) moved onto the GPU: each iteration yields a freshly generated (x, y, knobs) minibatch,
    `datapoints // batch_size` batches per epoch.  Nothing is stored.  For the comp_4c effects a minibatch is ONE kernel launch
    (st_synth_comp4c); measured rates: profiles/r03_train_loop_throughput.txt."""

    def __init__(self, dataset, batch_size, device="cuda:0", gen_windows=2048):
        """gen_windows: windows generated per call of the device generators (a few dozen small launches whatever the count:
        2048 at a time costs ~4.5 ms, i.e. 0.56 ms per 256-window minibatch -- below the fp32 step; 256 at a time costs 2.9 ms)."""
        self.ds, self.batch_size, self.device = dataset, int(batch_size), device
        self.per_call = max(1, int(gen_windows) // self.batch_size)

    def __len__(self):
        return self.ds.datapoints // self.batch_size

    def __iter__(self):
        """Chunks of `per_call` minibatches are generated on a SIDE stream one chunk ahead: the generator kernel (one lane per window in
        its sequential compressor stage) then runs beside the training steps of the previous chunk instead of in front of them."""
        import torch
        dev = torch.device(self.device)
        if dev.type != "cuda":
            left = len(self)
            while left > 0:
                k = min(self.per_call, left)
                x, y, kn = self.ds.batch_device(k * self.batch_size, self.device)
                for i in range(k):
                    sl = slice(i * self.batch_size, (i + 1) * self.batch_size)
                    yield x[sl], y[sl], kn[sl]
                left -= k
            return
        main = torch.cuda.current_stream(dev)
        side = getattr(self, "_side", None)
        if side is None:
            side = self._side = torch.cuda.Stream(device=dev)

        def produce(k):
            side.wait_stream(main)                       # buffers the caching allocator hands out may still be read by earlier steps
            with torch.cuda.stream(side):
                out = self.ds.batch_device(k * self.batch_size, dev)
                ev = torch.cuda.Event(); ev.record(side)
            return out, ev, k
        left = len(self)
        nxt = produce(min(self.per_call, left)) if left > 0 else None
        while nxt is not None:
            (x, y, kn), ev, k = nxt
            left -= k
            main.wait_event(ev)
            for t in (x, y, kn):
                t.record_stream(main)
            nxt = produce(min(self.per_call, left)) if left > 0 else None
            for i in range(k):
                sl = slice(i * self.batch_size, (i + 1) * self.batch_size)
                yield x[sl], y[sl], kn[sl]


class DeviceRecycledDataSet:
    """Device-resident counterpart of SynthAudioDataSet(recycle=True) (datasets.py:286-296): `datapoints` windows are
    generated ONCE -- input signals, knob settings and the effect all on the GPU (audio_device.py, st_compressor_4c) --
    and kept in HBM (x: datapoints x chunk floats, 6.5 GB for the reference's 200 000 x 8192); minibatches are then random
    index gathers on the device, so the training loop never waits for CPU workers.  iterate with batches(batch_size)."""

    def __init__(self, chunk_size, effect, sr=44100, datapoints=8000, y_size=None, augment=True, device="cuda:0", gen_batch=2048):
        import torch
        self.device = torch.device(device)
        self.datapoints, self.augment = int(datapoints), augment
        self.y_size = chunk_size if y_size is None else y_size
        gen = SynthAudioDataSet(chunk_size, effect, sr=sr, datapoints=datapoints, y_size=self.y_size, augment=False)
        self.x = torch.empty(self.datapoints, chunk_size, dtype=torch.float32, device=self.device)
        self.y = torch.empty(self.datapoints, self.y_size, dtype=torch.float32, device=self.device)
        self.knobs = torch.empty(self.datapoints, gen.num_knobs, dtype=torch.float32, device=self.device)
        for i in range(0, self.datapoints, gen_batch):
            n = min(gen_batch, self.datapoints - i)
            xb, yb, kb = gen.batch_device(n, self.device)
            self.x[i:i + n], self.y[i:i + n], self.knobs[i:i + n] = xb, yb, kb

    def __len__(self):
        return self.datapoints

    def batches(self, batch_size, shuffle=True):
        """One epoch of (x, y, knobs) device batches (drop_last); do_augment's random polarity flip per item."""
        import torch
        order = torch.randperm(self.datapoints, device=self.device) if shuffle else torch.arange(self.datapoints, device=self.device)
        for i in range(0, self.datapoints - batch_size + 1, batch_size):
            idx = order[i:i + batch_size]
            x, y, k = self.x[idx], self.y[idx], self.knobs[idx]
            if self.augment:
                sgn = torch.where(torch.rand(batch_size, 1, device=self.device) < 0.5, -1.0, 1.0)
                x, y = x * sgn, y * sgn
            yield x, y, k


# ------------------------------------------------------------------------------------------------ pre-recorded audio pairs
def parse_knob_string(name, ext=".wav", dtype=np.float32):
    """Knob settings from a target file name (datasets.py:178-186): everything after the first double underscore, e.g.
    'target_9400_Compressor_4c__-10.95__3.428__0.005043__0.01308.wav' -> [-10.95, 3.428, 0.005043, 0.01308] (world units)."""
    return np.array([float(v) for v in name.replace(ext, "").split("__")[1:]], dtype=dtype)


class AudioFileDataSet(Dataset):
    """Windows cut from pre-recorded input / target wav pairs -- the contract of signaltrain/datasets.py:64-259 (the feed of
    BASELINE configs[3], an LA2A recorded with 2-3 knobs): files `input_<n>_.wav` / `target_<n>_<effect>__k1__k2....wav` in
    `path`, sorted together; every item is a random `chunk_size` window of a random file pair, the target cropped to its last
    `y_size` samples, the knob settings parsed from the target's name and normalised to [-0.5, 0.5] with the effect's ranges,
    random polarity flip when `augment`.  Audio is preloaded (the reference's default).  `device_batches()` serves the same
    items from a device-resident copy of the audio as whole minibatches (gather by index on the GPU, no CPU workers)."""

    def __init__(self, chunk_size, effect, sr=44100, path="./Train/", datapoints=8000, dtype=np.float32, preload=True, rerun=False,
                 y_size=None, augment=True, align_end=True, view_of=None, compand=False):
        super().__init__()
        import glob
        import os
        if not preload or compand or view_of is not None:
            raise NotImplementedError("signaltrain_amd.AudioFileDataSet: only preload=True, compand=False, view_of=None are built")
        self.chunk_size, self.effect, self.sr, self.path, self.dtype = chunk_size, effect, sr, path, dtype
        self.datapoints, self.rerun_effect, self.augment, self.align_end = datapoints, rerun, augment, align_end
        self.y_size = chunk_size if y_size is None else y_size
        self.input_filenames = sorted(glob.glob(os.path.join(path, "input_*")))
        self.target_filenames = sorted(glob.glob(os.path.join(path, "target_*")))
        print("AudioFileDataSet: Found", len(self.input_filenames), "input files and", len(self.target_filenames), "target files in path", path)
        assert len(self.input_filenames) == len(self.target_filenames) and self.input_filenames, "input_* / target_* files must pair up"
        self.x, self.y, knobs = [], [], []
        for fi, ft in zip(self.input_filenames, self.target_filenames):
            a, b = audio.read_audio_file(fi, sr=sr)[0], audio.read_audio_file(ft, sr=sr)[0]
            if len(a) != len(b) and align_end:                     # some recordings are aligned at their ends only (datasets.py:128-136)
                n = min(len(a), len(b)); a, b = a[-n:], b[-n:]
            if getattr(effect, "is_inverse", False):
                a, b = b, a
            assert len(a) > chunk_size, f"{fi}: {len(a)} samples, need more than chunk_size = {chunk_size}"
            self.x.append(a.astype(dtype, copy=False)); self.y.append(b.astype(dtype, copy=False))
            knobs.append(parse_knob_string(os.path.basename(ft), dtype=dtype))
        self.knobs = np.stack(knobs)
        self.num_knobs = self.knobs.shape[1]
        assert self.num_knobs == len(effect.knob_ranges), "knob count in the file names must match the effect's knob_ranges"
        self._dev = None

    def __len__(self):
        return self.datapoints

    def knobs_nn(self, knobs_wc):
        kr = np.asarray(self.effect.knob_ranges, dtype=np.float64)
        return ((knobs_wc - kr[:, 0]) / (kr[:, 1] - kr[:, 0]) - 0.5).astype(self.dtype)

    def get_single_chunk(self):
        i = np.random.randint(0, high=len(self.x))
        xa, ya, kw = self.x[i], self.y[i], self.knobs[i]
        ibgn = np.random.randint(0, len(xa) - self.chunk_size)
        x_item, y_item = xa[ibgn:ibgn + self.chunk_size], ya[ibgn:ibgn + self.chunk_size]
        if self.rerun_effect:
            y_item, x_item = self.effect.go_wc(x_item, kw)
        y_item = y_item[-self.y_size:]
        if self.augment:
            x_item, y_item = do_augment(x_item, y_item)
        return x_item.astype(self.dtype, copy=False), y_item.astype(self.dtype, copy=False), self.knobs_nn(kw)

    def __getitem__(self, idx):
        return self.get_single_chunk()

    # ---- the same items as device minibatches
    def _to_device(self, device):
        import torch
        if self._dev is None or self._dev["x"].device != torch.device(device):
            off = np.cumsum([0] + [len(a) for a in self.x])
            self._dev = {"x": torch.from_numpy(np.concatenate(self.x)).to(device), "y": torch.from_numpy(np.concatenate(self.y)).to(device),
                         "off": torch.from_numpy(off[:-1]).to(device), "len": torch.tensor([len(a) for a in self.x], device=device),
                         "kn": torch.from_numpy(np.stack([self.knobs_nn(k) for k in self.knobs])).to(device)}
        return self._dev

    def batch_device(self, B, device="cuda:0"):
        """B items as device tensors: random file, random window start, gathered from the device-resident audio."""
        import torch
        if self.rerun_effect:
            raise NotImplementedError("rerun=True (target_type='chunk') re-runs a host effect per window: use the CPU loader")
        d = self._to_device(device)
        fi = torch.randint(0, len(self.x), (B,), device=device)
        start = d["off"][fi] + (torch.rand(B, device=device) * (d["len"][fi] - self.chunk_size).float()).long()
        ar = torch.arange(self.chunk_size, device=device)
        x = d["x"][start[:, None] + ar[None, :]]
        y = d["y"][start[:, None] + ar[None, self.chunk_size - self.y_size:]]
        if self.augment:
            sgn = torch.where(torch.rand(B, 1, device=device) < 0.5, -1.0, 1.0)
            x, y = x * sgn, y * sgn
        return x, y, d["kn"][fi]

    def device_batches(self, batch_size, device="cuda:0"):
        for _ in range(self.datapoints // batch_size):
            yield self.batch_device(batch_size, device)
