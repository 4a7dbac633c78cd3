"""One node, N ranks: start a script under ``torch.distributed.run`` as CHILD processes (one per GPU, rendezvous on 127.0.0.1).

Used by ``bench.py --gpus N`` and ``inference.py --gpus N`` so that the plain command works without a launcher in front of it.  Nothing is
exec'ed (a process that has touched the GPU must not replace itself on this pool): the children are started, their output relayed, their
exit code returned; a failing rank makes the launcher (max-restarts 0) and the caller exit non-zero.  The children run in a process group
of their own: when the launch outlives PYGIM_LAUNCH_TIMEOUT seconds that group -- by its id, nothing else -- is ended and 124 returned.
"""
import os
import signal
import socket
import subprocess
import sys


def self_launch(script, n_gpus, argv, threads_per_rank=1, tag="launch"):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, int(threads_per_rank))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--max-restarts", "0", script] + list(argv)
    limit = float(os.environ.get("PYGIM_LAUNCH_TIMEOUT", "2400"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"[{tag}] the {n_gpus}-rank launch did not finish within {limit:.0f} s: ending its process group", file=sys.stderr)
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        child.wait()
        return 124
