"""``SemanticMap`` -- RGB / semantic layers next to the elevation map (reference EM/semantic_map.py:16-400).

Same bookkeeping surface as the reference (layer_names, layer_specs_points, channel -> fusion regex mapping with a
"default", dynamic add_layer); the layers themselves live on the device inside the elevation map's context and all
fusions of one frame run in a single ``emap_semantic_update`` call."""
from __future__ import annotations

import ctypes as ct
import re
import sys
from typing import Dict, List

import numpy as np

from ._lib import EmapSemSpec, f32p
from .fusion.fusion_manager import FusionManager

_KIND = {"average": 0, "class_average": 1, "class_bayesian": 2, "bayesian_inference": 3}


class SemanticMap:
    def __init__(self, param, emap):
        self.param = param
        self._emap = emap
        self.layer_specs_points: Dict[str, str] = {}
        self.layer_specs_image: Dict[str, str] = {}
        self.layer_names: List[str] = []
        self.unique_fusion = list(param.fusion_algorithms)
        self.elements_to_shift = {}
        self.fusion_manager = FusionManager(param)
        self.initialize_fusion()

    # ---- reference surface ---------------------------------------------------------------------------
    def clear(self):
        if self.layer_names:
            self._emap._chk(self._emap._lib.emap_semantic_clear(self._emap._ctx))

    def initialize_fusion(self):
        for fusion in self.unique_fusion:
            self.fusion_manager.register_plugin(fusion)

    def update_fusion_setting(self):
        pass  # the reference re-flags persistent new_map layers here (:64-78); the alpha planes of the device store need no flag

    def add_layer(self, name):
        if name not in self.layer_names:
            self.layer_names.append(name)
            self._emap._chk(self._emap._lib.emap_semantic_configure(self._emap._ctx, len(self.layer_names)))

    def shift_map_xy(self, shift_value):
        pass  # done on the device together with the elevation map (emap_shift)

    def get_fusion(self, channels: List[str], channel_fusions: Dict[str, str], layer_specs: Dict[str, str]):
        """channel -> fusion algorithm (reference semantic_map.py:141-178)."""
        fusion_list, process_channels = [], []
        for channel in channels:
            if channel not in layer_specs:
                matched = self.get_matching_fusion(channel, channel_fusions)
                if matched is None:
                    if "default" in channel_fusions:
                        default_fusion = channel_fusions["default"]
                        print(f"[WARNING] Layer {channel} not found in layer_specs. Using {default_fusion} algorithm as default.", file=sys.stderr)
                        layer_specs[channel] = default_fusion
                        self.update_fusion_setting()
                    else:
                        print(f"[WARNING] Layer {channel} not found in layer_specs ({layer_specs}) and no default fusion is configured. Skipping.", file=sys.stderr)
                        continue
                else:
                    layer_specs[channel] = matched
                    self.update_fusion_setting()
            fusion_list.append(layer_specs[channel])
            process_channels.append(channel)
        return process_channels, fusion_list

    def get_matching_fusion(self, channel: str, fusion_algs: Dict[str, str]):
        for fusion_alg, alg_value in fusion_algs.items():
            if re.match(f"^{fusion_alg}$", channel):
                return alg_value
        return None

    def get_indices_fusion(self, pcl_channels: List[str], fusion_alg: str, layer_specs: Dict[str, str]):
        """column indices of the cloud and layer indices handled by ``fusion_alg`` (reference :196-221)."""
        pcl_val_list = [layer_specs[x] for x in pcl_channels]
        pcl_indices = np.array([idp + 3 for idp, x in enumerate(pcl_val_list) if x == fusion_alg], dtype=np.int32)
        layer_indices = np.array([self.layer_names.index(key) for key, val in layer_specs.items()
                                  if key in pcl_channels and val == fusion_alg], dtype=np.int32)
        return pcl_indices, layer_indices

    def prepare(self, channels):
        """called before the elevation frame: make sure every channel has a layer (and the count plane exists)."""
        process_channels, fusions = self.get_fusion(channels, self.param.pointcloud_channel_fusions, self.layer_specs_points)
        for channel in process_channels:
            if channel not in self.layer_names:
                print(f"Layer {channel} not found, adding it to the semantic map", file=sys.stderr)
                self.add_layer(channel)
        return process_channels, fusions

    def frame_fusions(self, channels):
        """what one frame's extra channels ask for: ``(spec, class_max_jobs)`` -- the ``emap_sem_spec`` of the sum / colour fusions (None
        when there is none) and the ``class_max`` fusions as ``(plugin, cloud columns, layer indices)``, which run as a pass of their
        own on the device (reference semantic_map.py:223-259: one ``plugin(...)`` call per fusion algorithm)."""
        process_channels, fusions = self.prepare(channels)
        if not process_channels:
            return None, []
        spec = EmapSemSpec()
        spec.alpha = float(self.param.average_weight)
        ns = nc = 0
        jobs = []
        for fusion in sorted(set(fusions)):
            plug = self.fusion_manager.get_plugin(fusion, "pointcloud")
            if plug is None:
                continue
            pcl_ids, layer_ids = self.get_indices_fusion(process_channels, fusion, self.layer_specs_points)
            if plug.kind == "class_max":      # a frame of its own on the device (per-layer maxima over exact class sums)
                jobs.append((plug, pcl_ids, layer_ids))
                continue
            for ch, ly in zip(pcl_ids, layer_ids):
                if plug.kind == "color":
                    if nc >= 4:
                        raise ValueError("at most 4 colour channels per cloud")
                    spec.col_chan[nc], spec.col_layer[nc] = int(ch), int(ly); nc += 1
                else:
                    if ns >= 16:
                        raise ValueError("at most 16 averaged channels per cloud")
                    spec.sum_chan[ns], spec.sum_layer[ns], spec.sum_kind[ns] = int(ch), int(ly), _KIND[plug.kind]; ns += 1
        spec.n_sum, spec.n_col = ns, nc
        return (spec if ns + nc else None), jobs

    def declare_frame(self, emap, channels):
        """before ``emap_update``: hand the frame its sum / colour fusions (``emap_frame_semantics``: they run INSIDE the frame, as
        ``update_layers_pointcloud`` does inside the reference's ``update_map_with_kernel``, elevation_mapping.py:368); returns the
        ``class_max`` jobs to run after it (``finish_frame``)."""
        spec, jobs = self.frame_fusions(channels)
        emap._chk(emap._lib.emap_frame_semantics(emap._ctx, ct.byref(spec) if spec is not None else None, 1 if jobs else 0))
        return jobs

    def finish_frame(self, emap, jobs, R, t):
        for plug, pcl_ids, layer_ids in jobs:
            plug.fuse(emap, pcl_ids, layer_ids, R, t)

    def update_layers_pointcloud(self, emap, channels, R, t):
        """fuse the extra channels of the bound cloud AFTER a frame that did not declare them (reference semantic_map.py:223-259;
        ``update_map_with_kernel`` itself goes through ``declare_frame`` / ``finish_frame``)."""
        spec, jobs = self.frame_fusions(channels)
        self.finish_frame(emap, jobs, R, t)
        if spec is None:
            return
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
        t = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
        emap._chk(emap._lib.emap_semantic_update(emap._ctx, f32p(R), f32p(t), ct.byref(spec)))

    def update_layers_image(self, emap, image, channels, image_height, image_width):
        """sample the image into the semantic layers through the correspondence computed by
        ``emap_image_correspondence`` (reference semantic_map.py:261-306)."""
        process_channels, fusion_methods = self.get_fusion(channels, self.param.image_channel_fusions, self.layer_specs_image)
        image = np.ascontiguousarray(image, np.float32)
        for j, (fusion, channel) in enumerate(zip(fusion_methods, process_channels)):
            if channel not in self.layer_names:
                print(f"Layer {channel} not found, adding it to the semantic map", file=sys.stderr)
                self.add_layer(channel)
            plug = self.fusion_manager.get_plugin(fusion, "image")
            if plug is None:
                continue
            idx = self.layer_names.index(channel)
            if plug.kind == "color":      # the reference hands the whole stack over and reads planes 0..2 as r, g, b
                img, kind = image, 1
            else:
                img, kind = image[j:j + 1], 0
            emap._chk(emap._lib.emap_image_fuse(emap._ctx, kind, idx, f32p(np.ascontiguousarray(img)), int(img.shape[0]),
                                                int(image_height), int(image_width), ct.c_double(getattr(plug, "alpha", 0.7))))

    # ---- read-back -----------------------------------------------------------------------------------
    def _layer(self, idx):
        out = np.empty((self._emap.rows, self._emap.cell_n), np.float32)
        self._emap._chk(self._emap._lib.emap_semantic_get_layer(self._emap._ctx, int(idx), f32p(out)))
        return out

    @property
    def semantic_map(self):
        if not self.layer_names:
            return np.zeros((0, self._emap.rows, self._emap.cell_n), np.float32)
        return np.stack([self._layer(i) for i in range(len(self.layer_names))], axis=0)

    def get_layer(self, name_or_idx):
        """one raw ``(rows, cell_n)`` layer (a 8192^2 map holds 268 MB per layer: read what you need, not the stack)"""
        return self._layer(self.layer_names.index(name_or_idx) if isinstance(name_or_idx, str) else int(name_or_idx))

    def set_layer(self, name_or_idx, array):
        idx = self.layer_names.index(name_or_idx) if isinstance(name_or_idx, str) else int(name_or_idx)
        a = np.ascontiguousarray(array, np.float32)
        self._emap._chk(self._emap._lib.emap_semantic_set_layer(self._emap._ctx, idx, f32p(a)))

    # the reference's persistent ``new_map`` layers of the class_bayesian fusion (semantic_map.py:54-56): Dirichlet pseudo-counts
    def get_alpha(self, name_or_idx):
        idx = self.layer_names.index(name_or_idx) if isinstance(name_or_idx, str) else int(name_or_idx)
        out = np.empty((self._emap.rows, self._emap.cell_n), np.float32)
        self._emap._chk(self._emap._lib.emap_semantic_get_alpha(self._emap._ctx, idx, f32p(out)))
        return out

    def set_alpha(self, name_or_idx, array):
        idx = self.layer_names.index(name_or_idx) if isinstance(name_or_idx, str) else int(name_or_idx)
        a = np.ascontiguousarray(array, np.float32)
        self._emap._chk(self._emap._lib.emap_semantic_set_alpha(self._emap._ctx, idx, f32p(a)))

    def get_id_max(self, name_or_idx):
        """class-id plane of a ``class_max`` layer (the reference's ``elements_to_shift["id_max"][layer]``, uint32; moves with the map)"""
        return self.get_alpha(name_or_idx).view(np.uint32)

    def get_index(self, name):
        return self.layer_names.index(name) if name in self.layer_names else -1

    def get_map_with_name(self, name):
        """border-stripped layer (reference :329-386; colour layers are the packed 0x00RRGGBB float plane)."""
        return self._layer(self.layer_names.index(name))[1:-1, 1:-1]

    get_rgb = get_map_with_name
    get_semantic = get_map_with_name

    # ---- the reference's small host helpers, kept for callers that use them directly ------------------------------------------
    def process_map_for_publish(self, input_map):
        """border-stripped copy of one layer (reference :376-386)"""
        return np.array(input_map, copy=True)[1:-1, 1:-1]

    def get_layer_indices(self, fusion_alg, layer_specs):
        """positions in ``layer_specs`` of the layers fused by ``fusion_alg`` (reference :184-197, its chained comparison
        ``key in val == fusion_alg`` included: a layer counts when its NAME is a substring of its fusion's name and that is the one asked for)"""
        return np.array([it for it, (key, val) in enumerate(layer_specs.items()) if key in val and val == fusion_alg], dtype=np.int32)

    def decode_max(self, mer):
        """float32 values that carry (half probability | class id << 16) -> (probability as float32, class id) (reference :311-327)"""
        bits = np.ascontiguousarray(mer, np.float32).view(np.uint32)
        prob = (bits & np.uint32(0xFFFF)).astype(np.uint16).view(np.float16).astype(np.float32)
        return prob, bits >> np.uint32(16)

    def pad_value(self, x, shift_value, idx=None, value=0.0):
        """fill the band a shift by ``shift_value`` (rows, columns) vacated in a (layers, rows, columns) stack (reference :99-125);
        host arrays only -- the device layers are padded by ``emap_shift`` itself"""
        sel = slice(None) if idx is None else idx
        if shift_value[0] > 0:
            x[sel, : shift_value[0], :] = value
        elif shift_value[0] < 0:
            x[sel, shift_value[0]:, :] = value
        if shift_value[1] > 0:
            x[sel, :, : shift_value[1]] = value
        elif shift_value[1] < 0:
            x[sel, :, shift_value[1]:] = value
