"""BASELINE configs[2] (Seal bbox distillation) on one GPU, phase by phase, for a kernel trace: the phases of bench.py's
seal_section separated by a marker kernel (`k_sph_from_ray`, which nothing on this path launches).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/seal_phases.py
    python tools/seal_phases.py --summarize <dir> <tag>      -> markdown on stdout"""
import csv, glob, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))

PHASES = ["teacher training (256 steps of the two-encoder net) + student set-up", "local pretraining: 8 epochs (one graph replay each)",
          "teacher proxy render: 8 batches of 4,096 rays", "fine-tuning warm-up (16 eager steps, capture, replays)",
          "fine-tuning: 32 steps incl. the proxy render", "target renders for the cached steps", "fine-tuning: 32 steps on cached targets"]
REPS = [1, 8, 8, 40, 32, 8, 32]
SHOWN = (1, 2, 4, 6)


def summarize(d, tag):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    phase, acc = 0, [dict() for _ in PHASES]
    spans = [[None, None] for _ in PHASES]
    for r in rows:
        if "k_sph_from_ray" in r["Kernel_Name"]:
            phase += 1
            continue
        if phase >= len(PHASES):
            break
        n = r["Kernel_Name"].replace("void ", "").replace("s3d::(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:70]
        a = acc[phase].setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        s = spans[phase]
        s[0] = int(r["Start_Timestamp"]) if s[0] is None else s[0]
        s[1] = int(r["End_Timestamp"])
    print(f"# rocprofv3 kernel trace `{tag}` — Seal-3D distillation (BASELINE configs[2]) phase by phase (`tools/seal_phases.py`, `tools/profile_seal.sh`)\n")
    print("bbox edit (translate 0.3), teacher + student two-encoder NGP nets (`nerf/network.py` on the fused path), `pretraining_local_point_step` 0.005 "
          "(7.5e5 lattice points, one chunk), 4,096 rays per fine-tuning step, HIP-graph replay.  Per unit = per epoch / batch / step.\n")
    for p in SHOWN:
        tot = sum(v[1] for v in acc[p].values())
        wall = (spans[p][1] - spans[p][0]) / 1e3 if spans[p][0] is not None else 0.0
        print(f"## {PHASES[p]}\n\nkernel time {tot / REPS[p]:.1f} us per unit, first-to-last kernel {wall / REPS[p]:.1f} us per unit, "
              f"{sum(v[0] for v in acc[p].values()) / REPS[p]:.1f} launches per unit\n")
        print("| kernel | launches/unit | us/launch | us/unit | % |")
        print("|---|---|---|---|---|")
        for n, (c, us) in sorted(acc[p].items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"| `{n}` | {c / REPS[p]:.1f} | {us / c:.1f} | {us / REPS[p]:.1f} | {100 * us / tot:.1f} |")
        print()


def main():
    import torch
    import bench, s3d_hip
    from nerf import network, synthetic as syn
    from nerf.trainer import GraphedTrainer
    from sealnerf import GraphedSealTrainer, SealBBoxMapper, make_student, make_teacher
    sys.argv = sys.argv[:1]
    args = bench.parse()
    dev = torch.device("cuda")
    R = s3d_hip.RaymarchingBackend
    grid, bits = syn.lego_like_density_grid(seed=0)
    batches, _ = bench.make_batches(32, args.num_rays, 0, dev, R, torch.from_numpy(bits).to(dev), syn.lego_like_boxes(0))
    mark_o = torch.zeros(1, 3, device=dev)
    mark_d = torch.ones(1, 3, device=dev)
    mark_c = torch.zeros(1, 2, device=dev)

    def mark():
        torch.cuda.synchronize()
        R.sph_from_ray(mark_o, mark_d, 1.0, 1, mark_c)
        torch.cuda.synchronize()
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    torch.manual_seed(17)
    teacher = make_teacher(network.NeRFNetwork, **kw).to(dev)
    ttr = GraphedTrainer(teacher, args.num_rays, lr=1e-2, fp16=True, update_extra_interval=16)
    for i in range(args.seal_teacher_steps):
        ttr.train_step(*batches[i % len(batches)])
    del ttr
    student = make_student(network.NeRFNetwork, **kw).to(dev)
    student.load_state_dict(teacher.state_dict())
    student.mean_count, student.mean_density, student.iter_density = teacher.mean_count, teacher.mean_density, teacher.iter_density
    mapper = SealBBoxMapper(bench.SEAL_BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    tr = GraphedSealTrainer(student, teacher, args.num_rays, lr=1e-2, fp16=True, update_extra_interval=16)
    n_local = tr.init_pretraining(batch_size=6144000, lr=0.05, local_point_step=0.005)
    tr.pretrain_one_epoch(); tr.pretrain_one_epoch()
    mark()
    for _ in range(REPS[1]):
        tr.pretrain_one_epoch()
    mark()
    for i in range(REPS[2]):
        tr.proxy_truth(batches[i][0], batches[i][1])
    mark()
    for i in range(REPS[3]):
        tr.train_step(batches[i % 32][0], batches[i % 32][1])
    mark()
    for i in range(REPS[4]):
        tr.train_step(batches[i % 32][0], batches[i % 32][1])
    mark()
    targets = [tr.proxy_truth(b[0], b[1]) for b in batches[:8]]
    mark()   # (the target renders land in a phase of their own: index 5 below is the cached steps)
    for i in range(REPS[6]):
        tr.train_step(batches[i % 8][0], batches[i % 8][1], *targets[i % 8])
    mark()
    print("local points", n_local)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        main()
