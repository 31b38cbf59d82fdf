"""`python -m lajolla_public_amd [-o output_file_name] [--spp N] [--gpus N | --devices 0,1,..] filename.xml ...` — the reference's
driver loop (main.cpp:12-51) on the HIP path: parse, render, write to `-o` or the scene's own output file name (PFM / EXR).
`-t num_threads` is accepted and ignored (the render does not run on host threads).  `--gpus N` renders on the first N devices of
the node from this one process (a device group: tiles sharded t % N, one RCCL reduce); `--devices` names them (ids may repeat:
logical ranks on one GPU)."""
import sys
import time

from . import Context, DeviceGroup, GroupScene, Scene, parse_scene, render, render_group, write_image


def main(argv):
    if not argv:
        print("[Usage] python -m lajolla_public_amd [-t num_threads] [-o output_file_name] [--spp N] [--gpus N | --devices i,j,..] filename.xml")
        return 0
    output, spp, filenames, devices = "", 0, [], None
    i = 0
    while i < len(argv):
        if argv[i] == "-t":
            i += 1
        elif argv[i] == "-o":
            i += 1
            output = argv[i]
        elif argv[i] == "--spp":
            i += 1
            spp = int(argv[i])
        elif argv[i] == "--gpus":
            i += 1
            devices = list(range(int(argv[i])))
        elif argv[i] == "--devices":
            i += 1
            devices = [int(d) for d in argv[i].split(",")]
        else:
            filenames.append(argv[i])
        i += 1
    group = DeviceGroup(devices) if devices and len(devices) > 1 else None
    ctx = None if group else Context(devices[0] if devices else 0)
    for filename in filenames:
        t0 = time.perf_counter()
        print(f"Parsing and constructing scene {filename}.")
        hs = parse_scene(filename)
        sc = GroupScene(group, hs) if group else Scene(ctx, hs)
        print(f"Done. Took {time.perf_counter() - t0:.6g} seconds.")
        print("Rendering...")
        t0 = time.perf_counter()
        img = render_group(sc, spp=spp) if group else render(sc, spp=spp)
        if output == "":
            output = hs.desc.output_filename.decode()
        print(f"Done. Took {time.perf_counter() - t0:.6g} seconds.")
        write_image(output, img)
        print(f"Image written to {output}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
