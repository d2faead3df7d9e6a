import os, sys, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
backend=sys.argv[1]
torch.cuda.set_device(0)
try:
    if backend=="nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda",0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    x=torch.arange(4,dtype=torch.float64,device="cuda")+10*rank
    y=torch.empty(4,dtype=torch.float64,device="cuda")
    try:
        dist.all_to_all_single(y,x,[2,2],[2,2])
        torch.cuda.synchronize(); print(backend,"rank",rank,"all_to_all ok",y.tolist(),flush=True)
    except Exception as e:
        print(backend,"rank",rank,"all_to_all_single failed:",repr(e)[:200],flush=True)
    try:
        ops=[dist.P2POp(dist.isend,x[:2],1-rank), dist.P2POp(dist.irecv,y[:2],1-rank)]
        for w in dist.batch_isend_irecv(ops): w.wait()
        torch.cuda.synchronize(); print(backend,"rank",rank,"p2p ok",y[:2].tolist(),flush=True)
    except Exception as e:
        print(backend,"rank",rank,"p2p failed:",repr(e)[:200],flush=True)
    t=torch.ones(1,dtype=torch.float64,device="cuda")*(rank+1); dist.all_reduce(t); print(backend,"rank",rank,"allreduce",t.item(),flush=True)
except Exception as e:
    print(backend,"rank",rank,"FAILED",repr(e)[:300],flush=True)
