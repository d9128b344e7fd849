#!/usr/bin/env python3
"""Applies the HIP binding to a SCRATCH COPY of the reference tree (never to /root/reference,
never into this repository): the edits INTEGRATION.md section B describes, as anchored
insertions. Each hunk names the reference construct it attaches to; the inserted text is ours.

    python apply_hip_binding.py <copy of the reference tree>
"""
import sys

HUNKS = {
    "smolscale-private.h": [
        # (anchor: a line of the reference that must exist exactly once, 'before'|'after', inserted text)
        # 1. three optional slots at the end of SmolImplementation (reference smolscale-private.h:270-278)
        ("    const SmolRepackMeta *repack_meta;\n", "after",
         "    /* Optional whole-range producer (a GPU backend): when init_batch_func() accepts the\n"
         "     * context, do_rows() calls batch_func() once per row range instead of looping. */\n"
         "    int (*batch_func) (const SmolScaleCtx *scale_ctx, void *outrows_dest,\n"
         "                       uint32_t first_row, uint32_t n_rows);\n"
         "    int (*init_batch_func) (SmolScaleCtx *scale_ctx);\n"
         "    void (*fini_batch_func) (SmolScaleCtx *scale_ctx);\n"),
        # 2. per-context state (reference struct SmolScaleCtx, smolscale-private.h:280-313)
        ("    unsigned int width_halvings, height_halvings;\n", "after",
         "\n"
         "    /* Set by get_implementations() when a whole-range producer took the context */\n"
         "    int (*batch_func) (const SmolScaleCtx *scale_ctx, void *outrows_dest,\n"
         "                       uint32_t first_row, uint32_t n_rows);\n"
         "    void (*fini_batch_func) (SmolScaleCtx *scale_ctx);\n"
         "    void *hip_plan;\n"),
        # 3. the getter, next to the AVX2 one (smolscale-private.h:331-333)
        ("const SmolImplementation *_smol_get_generic_implementation (void);\n", "after",
         "#ifdef SMOL_WITH_HIP\n"
         "const SmolImplementation *_smol_get_hip_implementation (void);\n"
         "#endif\n"),
    ],
    "smolscale.c": [
        # 4. do_rows(): hand the whole range over (reference smolscale.c:507-546). Anchor: the
        #    comment that opens the function's body after its declarations.
        ("    if (scale_ctx->storage_type == SMOL_STORAGE_128BPP)\n        n_parts_per_pixel = 2;\n", "before",
         "    if (scale_ctx->batch_func)\n"
         "    {\n"
         "        char *row = outrows_dest;\n"
         "\n"
         "        if (scale_ctx->batch_func (scale_ctx, outrows_dest, row_out_index, n_rows) != 0)\n"
         "            abort ();\n"
         "        /* post_row_func: once per row, ascending, on the packed row (smolscale.c:502-503) */\n"
         "        if (scale_ctx->post_row_func)\n"
         "            for (i = 0; i < n_rows; i++, row += scale_ctx->rowstride_out)\n"
         "                scale_ctx->post_row_func ((uint32_t *) row, scale_ctx->width_out, scale_ctx->user_data);\n"
         "        return;\n"
         "    }\n"
         "\n"),
        # 5. get_implementations(): ask the GPU backend last, after the CPU row functions are in
        #    place (smolscale.c:725-814) -- they remain the fall-through.
        ("    if (!scale_ctx->hfilter_func || !scale_ctx->vfilter_func)\n        abort ();\n", "after",
         "\n"
         "#ifdef SMOL_WITH_HIP\n"
         "    {\n"
         "        const SmolImplementation *hip = _smol_get_hip_implementation ();\n"
         "\n"
         "        if (hip->batch_func && hip->init_batch_func (scale_ctx))\n"
         "        {\n"
         "            scale_ctx->batch_func = hip->batch_func;\n"
         "            scale_ctx->fini_batch_func = hip->fini_batch_func;\n"
         "        }\n"
         "    }\n"
         "#endif\n"),
        # 6. smol_scale_init(): smol_scale_simple() uses an uninitialised stack context
        #    (smolscale.c:971), so the new fields are cleared here, before anything can read them.
        ("    scale_ctx->pixels_in = pixels_in;\n", "before",
         "    scale_ctx->batch_func = NULL;\n"
         "    scale_ctx->fini_batch_func = NULL;\n"
         "    scale_ctx->hip_plan = NULL;\n"),
        # 7. smol_scale_finalize() (smolscale.c:872-876)
        ("    free (scale_ctx->precalc_x_storage);\n", "before",
         "    if (scale_ctx->fini_batch_func)\n"
         "        scale_ctx->fini_batch_func (scale_ctx);\n"),
    ],
}


def main():
    root = sys.argv[1]
    for name, hunks in HUNKS.items():
        path = "%s/%s" % (root, name)
        text = open(path).read()
        for anchor, where, ins in hunks:
            if text.count(anchor) != 1:
                sys.exit("apply_hip_binding: anchor for %s found %d times in %s: %r"
                         % (where, text.count(anchor), name, anchor))
            text = text.replace(anchor, anchor + ins if where == "after" else ins + anchor)
        open(path, "w").write(text)
        print("patched", path)


if __name__ == "__main__":
    main()
