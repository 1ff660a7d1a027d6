"""Hunt for a kernel that is not bit-repeatable (tests/test_gpu_two_ranks.py eager-vs-graph mismatch on the driver's box).

  python tools/determinism_hunt.py run [--ranks 2] [--steps 20] [--replays 20] [--maxdisp 32 --H 128 --W 64] [--hog] [--nanfill] [--keep]

starts `--ranks` processes that share the GPU (python -m torch.distributed.run, gloo: exactly what the failing test does) and, in each,
  1. runs the data-parallel training step EAGERLY `--steps` times on identical inputs with every operator call traced
     (tests/op_trace.py) and compares each trace with the first one: the first differing entry names the operator;
  2. captures the step as a hipGraph and replays it `--replays` times, comparing the flat gradient with the eager step's, per parameter.
--hog adds one more process that keeps the GPU busy with unrelated kernels; --nanfill makes torch.empty return NaN-filled memory.
Each rank prints one summary line per finding; exit code 0 whether or not something was found (it is a measuring tool)."""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
  if p not in sys.path:
    sys.path.insert(0, p)


def hog(seconds):
  """Unrelated GPU work in a process of its own: large elementwise passes + matmuls, so that the waves of the process under test share
  CUs, LDS and HBM with somebody else's."""
  import torch
  a = torch.randn(4096, 4096, device='cuda')
  b = torch.randn(64 * 1024 * 1024, device='cuda')
  t0 = time.time()
  while time.time() - t0 < seconds:
    for _ in range(20):
      a = torch.tanh(a @ a * 1e-3)
      b = b * 1.0001 + 0.5
    torch.cuda.synchronize()


def worker(args):
  import torch
  import torch.distributed as dist
  import op_trace
  import two_rank_worker as trw
  rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
  if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  import recipe
  import models
  from mode_hip import data_parallel
  from mode_hip.graph_step import GraphedStep
  tag = '[rank %d]' % rank

  net = models.ModeDisparity(args.maxdisp, 'Sphere', args.H, args.W, 'Cassini').to(dev)
  net.load_state_dict(recipe.recipe_state_wc(recipe.load_manifest(), 77))
  net.train()
  reducer = data_parallel.GradAllReducer(net, fuse_accumulation=True)
  reducer.broadcast_parameters(net)
  left, right, gt = [t.to(dev) for t in trw.rank_batch(rank, args.maxdisp, args.H, args.W)]
  count = data_parallel.global_valid_count(~torch.isnan(gt))
  names_params = [(n, p) for n, p in net.named_parameters() if p.requires_grad]

  def body():
    reducer.zero_grad()
    loss = trw.step_loss(net, left, right, gt, count)
    loss.backward()
    return loss

  ctx = op_trace.nan_filled_allocations() if args.nanfill else None
  if ctx is not None:
    ctx.__enter__()

  # ---- 1. eager steps, traced
  ref_trace = ref_flat = ref_labels = None
  bad_steps = 0
  for s in range(args.steps):
    tr = op_trace.Trace(keep=args.keep, inputs=args.keep and bool(args.dump))
    with op_trace.tracing(tr):
      loss = body()
    torch.cuda.synchronize()
    t = tr.finish()
    flat = reducer.flat.cpu()
    if s == 0:
      ref_trace, ref_flat, ref_labels = t, flat, tr.labels
      nans = op_trace.nan_entries(tr.labels, t)
      print(tag, 'eager step 0: %d trace entries, loss %.9g, NaN entries: %s; NaNs in flat gradient: %d' % (
          len(tr.labels), float(loss), nans[:6], int(torch.isnan(flat).sum())), flush=True)
      continue
    assert tr.labels == ref_labels, 'the trace itself changed between steps'
    diff = op_trace.first_difference(tr.labels, ref_trace, t)
    if diff is not None or not torch.equal(flat, ref_flat):
      bad_steps += 1
      print(tag, 'EAGER step %d differs from step 0: first at entry %s' % (s, diff), flush=True)
      if args.dump and args.keep and diff is not None and bad_steps <= 4:
        i = diff[0]
        call = tr.labels[i].split('[')[0].rsplit('.in', 1)[0].rsplit('.saved', 1)[0]
        sel = [j for j, lb in enumerate(tr.labels) if lb.startswith(call + '[') or lb.startswith(call + '.in[') or lb.startswith(call + '.saved[')]
        os.makedirs(args.dump, exist_ok=True)
        path = os.path.join(args.dump, 'rank%d_step%d.pt' % (rank, s))
        torch.save({'first': tr.labels[i], 'labels': [tr.labels[j] for j in sel], 'ref': [ref_trace[j] for j in sel], 'cur': [t[j] for j in sel]}, path)
        print(tag, '  dumped', path, [tr.labels[j] for j in sel], flush=True)
      print(tag, '  flat gradient:', op_trace.param_report(names_params, ref_flat, flat), flush=True)
  print(tag, 'eager: %d of %d repeated steps differ from the first' % (bad_steps, args.steps - 1), flush=True)

  # ---- 2. hipGraph replays against the eager step
  # (no autograd graph of an eager step may be alive here: its AccumulateGrad nodes are bound to the default stream and a captured
  # backward that reuses them enqueues on that stream -- hipStreamEndCapture then dies with SIGSEGV instead of an error)
  del loss, tr
  if args.replays:
    gtr = op_trace.Trace(keep=args.keep)
    calls = {'n': 0}

    def fn_counted():  # GraphedStep calls it `warmup` times plainly, then once under capture: only that call is traced
      calls['n'] += 1
      if calls['n'] > 1 and args.trace_graph:
        with op_trace.tracing(gtr):
          return body()
      return body()

    graphed = GraphedStep(fn_counted, (left, right, gt, count), warmup=1)
    bad_replays = 0
    for r in range(args.replays):
      reducer.flat.fill_(float('nan'))
      graphed.replay()
      torch.cuda.synchronize()
      flat = reducer.flat.cpu()
      diff = None
      if args.trace_graph:
        assert gtr.labels == ref_labels, 'graph trace and eager trace list different operator calls'
        diff = op_trace.first_difference(gtr.labels, ref_trace, gtr.finish())
      if diff is not None or not torch.equal(flat, ref_flat):
        bad_replays += 1
        print(tag, 'GRAPH replay %d differs from eager step 0: first at entry %s' % (r, diff), flush=True)
        print(tag, '  flat gradient:', op_trace.param_report(names_params, ref_flat, flat, limit=40), flush=True)
    print(tag, 'graph: %d of %d replays differ from the eager step' % (bad_replays, args.replays), flush=True)
  if ctx is not None:
    ctx.__exit__(None, None, None)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


def free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('mode', choices=['run', 'worker', 'hog'])
  ap.add_argument('--ranks', type=int, default=2)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--replays', type=int, default=20)
  ap.add_argument('--maxdisp', type=int, default=32)
  ap.add_argument('--H', type=int, default=128)
  ap.add_argument('--W', type=int, default=64)
  ap.add_argument('--hog', action='store_true')
  ap.add_argument('--hog-seconds', type=float, default=600)
  ap.add_argument('--nanfill', action='store_true')
  ap.add_argument('--keep', action='store_true')
  ap.add_argument('--dump', default='', help='directory for the tensors of the first differing operator call (with --keep)')
  ap.add_argument('--trace-graph', action='store_true', help='record the operator outputs inside the captured step too')
  args = ap.parse_args()
  if args.mode == 'hog':
    return hog(args.hog_seconds)
  if args.mode == 'worker':
    return worker(args)
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  env['HSA_ENABLE_IPC_MODE_LEGACY'] = env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  hogp = None
  if args.hog:
    hogp = subprocess.Popen([sys.executable, os.path.abspath(__file__), 'hog', '--hog-seconds', str(args.hog_seconds)], env=env)
    time.sleep(8)
  fwd = ['--steps', str(args.steps), '--replays', str(args.replays), '--maxdisp', str(args.maxdisp), '--H', str(args.H), '--W', str(args.W)]
  fwd += (['--nanfill'] if args.nanfill else []) + (['--keep'] if args.keep else []) + (['--trace-graph'] if args.trace_graph else []) + (['--dump', args.dump] if args.dump else [])
  if args.ranks > 1:
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__), 'worker'] + fwd
  else:
    cmd = [sys.executable, os.path.abspath(__file__), 'worker'] + fwd
  try:
    r = subprocess.run(cmd, env=env, timeout=3000)
    print('workers exit code', r.returncode)
  finally:
    if hogp is not None:
      hogp.kill()
      hogp.wait()


if __name__ == '__main__':
  main()
