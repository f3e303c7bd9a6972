"""Helpers shared by the parity tests."""
import hashlib

import bp_oracle as o


def digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def pt_from_json(j):
    return None if j is None else (int(j[0], 16), int(j[1], 16))


def xyz_to_affine(xyz, c):
    """normalised Jacobian (12 u64, Montgomery) -> affine tuple / None"""
    x, y, z = [o.from_mont(o.limbs_to_int(xyz[4 * i:4 * i + 4]), c.p) for i in range(3)]
    if z == 0:
        assert x == 1 and y == 1
        return None
    assert z == 1, "result is not normalised"
    return (x, y)
