"""Stand-in: the geometry functions other modules bind by name at import (host arithmetic: the oracle's)."""
import torch

from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)

from .procrustes import align_rigid


def sample_image_grid(shape, device=torch.device("cpu")):
    return orc.pixel_grid(tuple(shape), device)


def homogenize_points(points):
    return orc.append_one(points)


def homogenize_vectors(vectors):
    return orc.append_zero(vectors)


def transform_rigid(homogeneous_coordinates, transformation):
    return orc.matvec(transformation, homogeneous_coordinates)


def unproject(coordinates, z, intrinsics):
    return orc.lift(coordinates, z, intrinsics)


def project_camera_space(points, intrinsics, epsilon=1e-5, infinity=1e8):
    return orc.pinhole(points, intrinsics, epsilon, infinity)


def project(points, extrinsics, intrinsics, epsilon=1e-5):
    return orc.world_to_image(points, extrinsics, intrinsics, epsilon)


def reproject_points(xyz, relative_transformations, intrinsics):
    return orc.warp_points(xyz, relative_transformations, intrinsics)


def compute_forward_flow(surfaces, extrinsics, intrinsics):
    return orc.forward_flow_positions(surfaces, extrinsics, intrinsics)


def compute_backward_flow(surfaces, extrinsics, intrinsics):
    return orc.backward_flow_positions(surfaces, extrinsics, intrinsics)


def get_extrinsics(inverse_relative_transformations):
    return orc.chain_poses(inverse_relative_transformations)


def align_surfaces(surfaces, backward_flows, backward_weights, indices):
    b, f, h, w, _ = surfaces.shape
    xy, _ = sample_image_grid((h, w), surfaces.device)
    later = surfaces[:, 1:].reshape(b, f - 1, h * w, 3)[:, :, indices]
    where = (xy + backward_flows).reshape(b, f - 1, h * w, 2)[:, :, indices]
    earlier = orc.bilinear_border(surfaces[:, :-1], where)
    weights = backward_weights.reshape(b, f - 1, h * w)[..., indices]
    return get_extrinsics(align_rigid(later, earlier, weights))  # (through the names bound HERE: install() rebinds them)


def compute_track_flow(surfaces, extrinsics, intrinsics, tracks):
    return orc.track_positions(surfaces, extrinsics, intrinsics, tracks)
