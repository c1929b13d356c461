"""Batch producer for the hot path: the semantics of reference datasets.py (Fuel pipeline) without
Fuel / HDF5, plus the pinned-host asynchronous loader the north star asks for.

Kept from the reference (datasets.py:206-298 and the Fuel transformers it chains):
  shuffle (train) / sequential order -> Batch(batch*sorting_mult) -> sort by length -> Unpack ->
  Batch(batch) -> drop short batches -> Padding (adds *_mask) -> time-major transpose of features /
  features_mask -> 80-sample chunking + per-batch quantisation of raw audio (datasets.py:194-203) ->
  SegmentSequence(seq_size+1, share_value=1, return_last=False, add_flag=True) TBPTT windows with
  start_flag (datasets.py:41-138, 286-292) -> optional feedback_noise_level source.

Replaced: the HDF5 VoiceData source (no dataset / h5py here) by a seeded synthetic source with the
same per-example fields, and the host-side numpy quantiser by the HIP kernel (parrot_amd.ops).
"""
from __future__ import annotations

import queue
import threading

import numpy

VOICES = ['arctic', 'blizzard', 'dimex', 'librispeech', 'pavoque', 'vctk']  # datasets.py:178-179


def _chunk(data, frame_size=80, axis=1):
    """datasets.py:27-28."""
    return numpy.stack(numpy.split(data, data.shape[axis] // frame_size, axis))


class VoiceData:
    """Synthetic stand-in for datasets.VoiceData (H5PYDataset): per-example variable-length
    (features [T,63], labels [U], raw_audio [T*80], speaker_index [1])."""

    def __init__(self, voice, which_sets=('train',), num_examples=64, seed=1234, min_frames=40,
                 max_frames=120, min_labels=5, max_labels=15, num_characters=43, num_speakers=21,
                 output_dim=63, frame_size=80):
        assert voice in VOICES
        self.voice, self.which_sets = voice, which_sets
        rng = numpy.random.RandomState(seed + sum(map(ord, ''.join(which_sets))))
        self.examples = []
        for _ in range(num_examples):
            T = int(rng.randint(min_frames, max_frames + 1))
            U = int(rng.randint(min_labels, max_labels + 1))
            self.examples.append(dict(
                features=rng.randn(T, output_dim).astype('float32'),
                labels=rng.randint(0, num_characters, size=(U,)).astype('int32'),
                raw_audio=rng.randn(T * frame_size).astype('float32'),
                speaker_index=rng.randint(0, num_speakers, size=(1,)).astype('int32')))
        self.num_examples = num_examples


class SegmentSequence:
    """datasets.py:41-138 -- cuts the time-major sources of each padded batch into windows.

    seq_size: window length; share_value: overlap; return_last=False drops a trailing window shorter
    than min_size + seq_size; add_flag appends start_flag (1 on the first window of a batch)."""

    def __init__(self, batches, sources, seq_size=100, which_sources=None, add_flag=False,
                 flag_name=None, min_size=10, return_last=True, share_value=0):
        self.batches, self._sources = batches, tuple(sources)
        self.which_sources = tuple(which_sources) if which_sources is not None else self._sources
        self.seq_size, self.add_flag, self.share_value = seq_size, add_flag, share_value
        self.min_size = min_size + (0 if return_last else seq_size)
        self.flag_name = flag_name or 'start_flag'

    @property
    def sources(self):
        return self._sources + ((self.flag_name,) if self.add_flag else ())

    def __iter__(self):
        for data in self.batches:
            idx0 = self._sources.index(self.which_sources[0])
            len_data = data[idx0].shape[0]
            step, flag = 0, 1
            while True:
                seg = list(data)
                for s in self.which_sources:
                    i = self._sources.index(s)
                    seg[i] = data[i][step:step + self.seq_size]
                step += self.seq_size
                step -= self.share_value
                last = step + self.min_size >= len_data
                if self.add_flag:
                    seg.append(flag)
                yield tuple(seg)
                flag = 0
                if last:
                    break


def get_raw_transformer(q_type, q_level, quantizer=None):
    """datasets.py:194-203: [T,B,80] float -> per-row (utterance) min-max + quantise -> [T,B,80] ints."""
    if quantizer is None:
        def quantizer(x, q_level, q_type):
            import torch
            from . import ops
            return ops.batch_quantize(torch.from_numpy(numpy.ascontiguousarray(x, dtype='float32')).cuda(),
                                      q_level, q_type).cpu().numpy()

    def transformer(batch):
        shp = batch.shape
        flat = batch.transpose(1, 0, 2).reshape((shp[1], -1))
        q = quantizer(flat, q_level, q_type)
        return q.reshape((shp[1], -1, 80)).transpose(1, 0, 2)
    return transformer


class DataStream:
    def __init__(self, sources, make_iter):
        self.sources, self._make_iter = tuple(sources), make_iter

    def get_epoch_iterator(self, as_dict=False):
        it = self._make_iter()
        if as_dict:
            return (dict(zip(self.sources, x)) for x in it)
        return it


def _pad(seqs, dtype, n=None):
    n = max(len(s) for s in seqs) if n is None else n
    out = numpy.zeros((len(seqs), n) + seqs[0].shape[1:], dtype=dtype)
    mask = numpy.zeros((len(seqs), n), dtype='float32')
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
        mask[i, :len(s)] = 1.
    return out, mask


def parrot_stream(voice, use_speaker=False, which_sets=('train',), batch_size=32, seq_size=50,
                  num_examples=None, sorting_mult=4, noise_level=None, labels_type='full_labels',
                  check_ratio=False, raw_data=True, q_type='mu-law', q_level=256, dataset=None,
                  quantizer=None, seed=1234, shard=None):
    """datasets.parrot_stream (datasets.py:206-298).  Returns a DataStream whose epoch iterator yields
    tuples in `sources` order: features [S,B,63], features_mask [S,B], [raw_audio [S,B,80]], labels
    [B,U], [labels_mask [B,U]], [speaker_index [B,1]], start_flag, [feedback_noise_level].

    shard = (rank, world): data-parallel producer.  Batches are formed exactly as for one process (same shuffle,
    same length-bucket sort, same drop rule on the GLOBAL batch) but only this rank's contiguous rows are
    padded / quantised / materialised; padding lengths are the global batch's, so every rank cuts the same
    number of TBPTT windows (= issues the same number of gradient all-reduces)."""
    assert labels_type in ['full_labels', 'phonemes', 'unconditional', 'unaligned_phonemes', 'text']
    if labels_type in ('full_labels', 'phonemes'):
        raise NotImplementedError(
            "frame-aligned labels are incompatible with Parrot's imatrix labels (model.py:511, SURVEY 8a)")
    if dataset is None:
        dataset = VoiceData(voice, which_sets, num_examples=num_examples or 64, seed=seed)
    n = num_examples or dataset.num_examples
    sorting_size = batch_size * sorting_mult
    sources = ['features', 'features_mask']
    if raw_data:
        sources.append('raw_audio')
    if labels_type != 'unconditional':
        sources += ['labels', 'labels_mask']
    if use_speaker:
        sources.append('speaker_index')
    raw_tf = get_raw_transformer(q_type, q_level, quantizer) if raw_data else None
    epoch = {'n': 0}

    def padded_batches():
        rng = numpy.random.RandomState(seed + epoch['n'])
        epoch['n'] += 1
        order = rng.permutation(n) if 'train' in which_sets else numpy.arange(n)
        exs = [dataset.examples[i] for i in order]
        if check_ratio and labels_type in ['unaligned_phonemes', 'text']:
            lo, hi = (8, 16) if labels_type == 'text' else (12., 25.)
            exs = [e for e in exs if lo <= len(e['features']) / float(len(e['labels'])) <= hi]
        for s in range(0, len(exs), sorting_size):
            chunk = sorted(exs[s:s + sorting_size], key=lambda e: len(e['features']))
            for b in range(0, len(chunk), batch_size):
                batch = chunk[b:b + batch_size]
                if len(batch) != batch_size:  # Filter(_check_batch_size)
                    continue
                t_max = max(len(e['features']) for e in batch)
                u_max = max(len(e['labels']) for e in batch)
                if shard is not None:
                    from .dist import shard_batch
                    lo, hi = shard_batch(batch_size, shard[0], shard[1])
                    batch = batch[lo:hi]
                feats, fmask = _pad([e['features'] for e in batch], 'float32', t_max)
                out = [feats.swapaxes(0, 1), fmask.swapaxes(0, 1)]  # time-major (datasets.py:274-275)
                if raw_data:
                    raw, _ = _pad([e['raw_audio'] for e in batch], 'float32', t_max * 80)
                    out.append(raw_tf(_chunk(raw)))
                if labels_type != 'unconditional':
                    lab, lmask = _pad([e['labels'] for e in batch], 'int32', u_max)
                    out += [lab, lmask]
                if use_speaker:
                    out.append(numpy.stack([e['speaker_index'] for e in batch]))
                yield tuple(out)

    seg_sources = ('features', 'features_mask') + (('raw_audio',) if raw_data else ())

    def make_iter():
        seg = SegmentSequence(padded_batches(), sources, seq_size=seq_size + 1, share_value=1,
                              return_last=False, add_flag=True, which_sources=seg_sources)
        for item in seg:
            yield item + ((noise_level,) if noise_level is not None else ())

    final_sources = tuple(sources) + ('start_flag',) + (('feedback_noise_level',) if noise_level is not None else ())
    return DataStream(final_sources, make_iter)


class PinnedAsyncLoader:
    """Double-buffered host->device feeder: a background thread pulls numpy batches from a stream, stages
    them in a RING of pinned host buffers (depth + 2 slots, reused; a slot is rewritten only after the H2D
    copies issued from it have completed) and the consumer issues non-blocking copies on a side HIP stream, so
    the next window's H2D transfer overlaps the current window's scan."""

    def __init__(self, stream: DataStream, device, depth=2):
        import torch
        self.stream, self.device, self.depth = stream, torch.device(device), depth
        self.sources = stream.sources
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._slots = [dict(bufs={}, event=None) for _ in range(depth + 2)]
        self.pinned_allocations = 0  # how many pinned buffers were ever created (tests: stays bounded)

    def _stage(self, slot, i, x):
        import torch
        if self._copy_stream is None:
            return torch.from_numpy(numpy.ascontiguousarray(x))
        src = torch.from_numpy(numpy.ascontiguousarray(x))
        key = (i, src.dtype)
        buf = slot['bufs'].get(key)
        if buf is None or buf.numel() < src.numel():
            buf = torch.empty(max(src.numel(), 1), dtype=src.dtype).pin_memory()
            slot['bufs'][key] = buf
            self.pinned_allocations += 1
        view = buf[:src.numel()].view(src.shape)
        view.copy_(src)
        return view

    def __iter__(self):
        import torch
        q: queue.Queue = queue.Queue(maxsize=self.depth)
        sentinel = object()

        stop = threading.Event()
        failure = []

        def put(x):  # a consumer that left early (break / exception) must not leave the worker blocked forever
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        slots = self._slots  # this iteration's ring: a straggling worker of an earlier iteration keeps ITS ring

        def worker():
            try:
                k = 0
                for item in self.stream.get_epoch_iterator():
                    if stop.is_set():
                        return
                    slot = slots[k % len(slots)]
                    k += 1
                    if slot['event'] is not None:
                        slot['event'].synchronize()  # copies issued from this slot's buffers are done
                    staged = [self._stage(slot, i, x) if isinstance(x, numpy.ndarray) else x
                              for i, x in enumerate(item)]
                    if not put((slot, staged)):
                        return
            except BaseException as e:  # re-raised in the consumer: an error is not an end of epoch
                failure.append(e)
            finally:
                put(sentinel)

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        try:
            yield from self._consume(q, sentinel, failure)
        finally:
            stop.set()
            while True:  # let a worker blocked in put() see the flag, then drop what it staged
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            th.join(timeout=5.0)
            if th.is_alive():  # still inside an event wait or a slow source: the next iteration gets a ring of its own
                self._slots = [dict(bufs={}, event=None) for _ in range(len(slots))]

    def _consume(self, q, sentinel, failure):
        import torch
        while True:
            got = q.get()
            if got is sentinel:
                if failure:
                    raise failure[0]
                break
            slot, staged = got
            out, ev = [], None
            if self._copy_stream is not None:
                with torch.cuda.stream(self._copy_stream):
                    for x in staged:
                        out.append(x.to(self.device, non_blocking=True) if isinstance(x, torch.Tensor) else x)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                slot['event'] = ev
                torch.cuda.current_stream(self.device).wait_event(ev)
                for x in out:  # the consumer stream now owns the buffers
                    if isinstance(x, torch.Tensor):
                        x.record_stream(torch.cuda.current_stream(self.device))
            else:
                out = staged
            yield dict(zip(self.sources, out))
