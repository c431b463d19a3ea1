"""Point-cloud fusion plugins of the MI355X backend (reference: EM/fusion/).  A plugin here is a *descriptor*: the
arithmetic of all selected fusions runs fused in one device call (``emap_semantic_update``), see fusion_manager."""
