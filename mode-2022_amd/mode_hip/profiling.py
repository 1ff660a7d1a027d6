"""Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline line).

Disabled by default (zero overhead: one attribute test per call).  When enabled, every native call made through
``mode_hip.functional`` is bracketed by two events recorded on the stream the kernel is launched on, together
with the ALGORITHMIC bytes / flops of that launch (see DESIGN.md for the per-unit figures)."""
import collections

import torch

ENABLED = False
_records = []  # (name, start_event, end_event, bytes, flops)


def enable(flag=True):
  global ENABLED
  ENABLED = flag
  _records.clear()


class _Region(object):
  __slots__ = ('name', 'nbytes', 'flops', 'start', 'stream')

  def __init__(self, name, nbytes, flops, device):
    self.name, self.nbytes, self.flops = name, nbytes, flops
    self.stream = torch.cuda.current_stream(device)

  def __enter__(self):
    self.start = torch.cuda.Event(enable_timing=True)
    self.start.record(self.stream)
    return self

  def __exit__(self, *exc):
    end = torch.cuda.Event(enable_timing=True)
    end.record(self.stream)
    _records.append((self.name, self.start, end, self.nbytes, self.flops))
    return False


class _Null(object):

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False


_NULL = _Null()


def region(name, nbytes=0, flops=0, device=None):
  return _Region(name, nbytes, flops, device) if ENABLED else _NULL


def summary():
  """{kernel: dict(calls, total_ms, avg_ms, bytes_per_call, flops_per_call, GBps, TFLOPs)}; call after a device sync."""
  agg = collections.OrderedDict()
  for name, s, e, nb, fl in _records:
    a = agg.setdefault(name, dict(calls=0, total_ms=0.0, bytes=0, flops=0))
    a['calls'] += 1
    a['total_ms'] += s.elapsed_time(e)
    a['bytes'] += nb
    a['flops'] += fl
  for a in agg.values():
    a['avg_ms'] = a['total_ms'] / a['calls']
    sec = a['total_ms'] * 1e-3
    a['GBps'] = a['bytes'] / sec / 1e9 if sec > 0 else 0.0
    a['TFLOPs'] = a['flops'] / sec / 1e12 if sec > 0 else 0.0
    a['bytes_per_call'] = a['bytes'] / a['calls']
    a['flops_per_call'] = a['flops'] / a['calls']
  return agg
