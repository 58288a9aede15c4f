"""bench.py's collective pattern on real RCCL with ONE rank (the box has one GPU; RCCL refuses two ranks on a device):
`torch.distributed` backend "nccl", the engine's stream wrapped as an ExternalStream, the positions packed behind the
step, the all-gather on a stream of its own behind an event, the next step enqueued beside it.  What cannot be
checked here is a transfer; what can is that every call of the pattern is accepted by the RCCL build of this image
and that the gathered block is the shard's own positions in the reference's order."""
import os
import socket

import numpy as np
import pytest

import cases
import ltrans_amd

pytestmark = pytest.mark.gpu


def test_all_gather_into_tensor_behind_the_step_one_rank():
    import torch
    import torch.distributed as dist
    from ltrans_amd.driver import LtransRun
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", device_id=dev, world_size=1, rank=0)
    try:
        case = cases.build_case("tiny_adv")
        e = cases.make_engine(case, ltrans_amd.Engine, {})
        run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
        run.init_hydro()
        n = case["cfg"].numpar
        send = torch.zeros(3 * n, dtype=torch.float64, device=dev)
        recv = torch.empty(3 * n, dtype=torch.float64, device=dev)
        gather_stream = torch.cuda.Stream(device=dev)
        eng_stream = torch.cuda.ExternalStream(e.stream(), device=dev)
        run.run(1, max_internal=2)
        want = e.get_particles(fields=("par",))["par"][:, :3].copy()
        e.pack_positions(send.data_ptr())
        packed = torch.cuda.Event()
        packed.record(eng_stream)
        gather_stream.wait_event(packed)
        with torch.cuda.stream(gather_stream):
            dist.all_gather_into_tensor(recv, send)
            done = torch.cuda.Event()
            done.record(gather_stream)
        run.run_external_timestep(2, max_internal=1)      # the next step runs beside the collective
        done.synchronize()
        e.sync()
        assert np.array_equal(recv.cpu().numpy().reshape(3, -1).T, want)
        # the job-wide reductions bench.py does on the device with this backend
        sw = torch.tensor([1], dtype=torch.int64, device=dev)
        dist.all_reduce(sw, op=dist.ReduceOp.MIN)
        assert int(sw.item()) == 1
        dist.barrier()
        e.close()
    finally:
        dist.destroy_process_group()
