"""dev helper: gnnb.hip with its local includes inlined and the public header addressed absolutely -- the text the ablation
builders (mk_abl*.py, mk_toptime.py) patch by marker and compile from /tmp."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gnn_branching_amd", "csrc")


def flat_source():
    def inline(text, seen):
        def sub(m):
            name = m.group(1)
            path = os.path.join(CSRC, name)
            if name.startswith("..") or not os.path.exists(path):
                return m.group(0)
            if name in seen:
                return ""
            seen.add(name)
            body = open(path).read().replace("#pragma once", "")
            return inline(body, seen)
        return re.sub(r'^#include "([^"]+)"$', sub, text, flags=re.M)
    src = inline(open(os.path.join(CSRC, "gnnb.hip")).read(), set())
    return src.replace('"../../include/gnnb.h"', '"' + os.path.abspath(os.path.join(CSRC, "..", "..", "include", "gnnb.h")) + '"')
