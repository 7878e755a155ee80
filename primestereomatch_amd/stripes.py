"""Row-stripe sharding of the hot path between ranks (bench.py --shard rows; C ABI: psm_set_rows / psm_set_map_buffer).

Rank g of G owns the output rows [g*R, min(H, (g+1)*R)) with R = ceil(H / G) of BOTH disparity maps, computed from all D
slices of both volumes - DispSel::CVSelect (src/DispSel.cpp:96-104) finished locally, nothing of the cost volumes or their
minima leaves the rank.  The one exchange per frame is an all-gather of the finished rows: because the stripes are aligned at
multiples of R, the gathered [G][2][R][W] tensor is the two whole maps up to a transpose of the first two axes.
The functions here are plain torch tensor code shared by bench.py (RCCL) and tests/test_dist_gloo.py (gloo, CPU)."""


def stripe_bounds(H: int, parts: int, rank: int):
    """(R, y0, y1): rows per stripe and this rank's rows [y0, y1) - empty (y0 == y1 == H) for ranks past the image."""
    R = -(-H // parts)
    y0 = min(H, rank * R)
    return R, y0, min(H, y0 + R)


def pack_stripe(maps, y0: int, y1: int, send, H: int, W: int, R: int):
    """maps: flat uint8 tensor holding [2][H][W]; send: flat uint8 tensor of 2*R*W -> the stripe rows of both maps."""
    send.view(2, R, W)[:, :y1 - y0].copy_(maps[:2 * H * W].view(2, H, W)[:, y0:y1])
    return send


def assemble(recv, world: int, H: int, W: int, R: int, out):
    """recv: flat uint8 tensor [world][2][R][W] as all_gather_into_tensor delivers it; out: flat uint8 tensor, [2][H][W] written."""
    src = recv.view(world, 2, R, W).permute(1, 0, 2, 3)            # [side][rank][row][x]
    if world * R == H:
        out[:2 * H * W].view(2, world, R, W).copy_(src)
    else:
        out[:2 * H * W].view(2, H, W).copy_(src.reshape(2, world * R, W)[:, :H])
    return out
