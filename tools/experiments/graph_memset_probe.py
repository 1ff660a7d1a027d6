"""Does a hipMemsetAsync captured into a hipGraph run when the graph is replayed?  (Round 4: 3/4 of the stride-2 1x1 input gradient --
zero-filled by hipMemsetAsync inside mode_conv1x1_bwd_data -- came back as garbage from the second replay of a captured step on.)

  python tools/experiments/graph_memset_probe.py"""
import ctypes

import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.restype = ctypes.c_int


def probe(name, nbytes, fill):
  dev = torch.device('cuda', 0)
  buf = torch.full((nbytes // 4,), 7.0, device=dev)
  side = torch.cuda.Stream(dev)
  g = torch.cuda.CUDAGraph()
  torch.cuda.synchronize()
  with torch.cuda.graph(g, stream=side):
    fill(buf)
    buf[:1] += 1.0  # a kernel node after the memset node
  torch.cuda.synchronize()
  res = []
  for r in range(3):
    buf.fill_(7.0)
    g.replay()
    torch.cuda.synchronize()
    res.append((float(buf[0]), int((buf[1:] != 0).sum())))
  ok = all(a == 1.0 and b == 0 for a, b in res)
  print('%-34s %10d bytes: %s   (buf[0], non-zero elements after it) per replay: %s' % (name, nbytes, 'ok' if ok else 'NOT ZEROED', res), flush=True)


def main():
  for nbytes in (4096, 1 << 20, 1 << 21, (1 << 21) + 4096, 64 << 20):
    probe('torch zero_()', nbytes, lambda b: b.zero_())
    probe('hipMemsetAsync', nbytes, lambda b: hip.hipMemsetAsync(b.data_ptr(), 0, b.numel() * 4, torch.cuda.current_stream().cuda_stream))
    probe('hipMemsetD32Async', nbytes, lambda b: hip.hipMemsetD32Async(b.data_ptr(), 0, b.numel(), torch.cuda.current_stream().cuda_stream))
  # the same memset between two kernels that use the buffer (the shape of mode_conv1x1_bwd_data inside a step)
  dev = torch.device('cuda', 0)
  a = torch.ones(1 << 18, device=dev)
  out = torch.empty(1 << 20, device=dev)
  side = torch.cuda.Stream(dev)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=side):
    out.copy_(torch.arange(1 << 20, device=dev, dtype=torch.float32))  # garbage from an earlier user of the block
    hip.hipMemsetAsync(out.data_ptr(), 0, out.numel() * 4, torch.cuda.current_stream().cuda_stream)
    out[::4] += a
  for r in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('memset between two kernels, replay %d: strided ones %d of %d, non-zero elsewhere %d' % (
        r, int((out[::4] == 1).sum()), out.numel() // 4, int((out.view(-1, 4)[:, 1:] != 0).sum())), flush=True)


if __name__ == '__main__':
  main()
