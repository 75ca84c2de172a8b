#!/bin/bash
# Does every patch under tools/experiments still apply to the commit PATCHES.txt names for it?  (the
# patches are the code of closed experiments; after a consolidation of the sources they stop applying to HEAD, so each is
# pinned to the newest commit whose tree takes it.)  Exit 1 on the first patch that does not apply, or that is not indexed.
set -u
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
cd "$root"
fail=0
indexed=""
while read -r patch commit strip numbers; do
  case "$patch" in ""|\#*) continue ;; esac
  indexed="$indexed $patch"
  [ -f "$here/$patch" ] || { echo "MISSING  $patch"; fail=1; continue; }
  [ -f "$root/$numbers" ] || { echo "NO-NUMBERS $patch ($numbers)"; fail=1; }
  d="$(mktemp -d)"
  if [ "$commit" = HEAD ]; then
    # the working tree as it stands (tracked files)
    git ls-files -z seigen_amd/csrc include bench.py tests tools seigen_amd/*.py | xargs -0 cp --parents -t "$d" 2>/dev/null
  else
    git archive "$commit" seigen_amd/csrc include bench.py tests tools $(git ls-tree --name-only "$commit" seigen_amd/ | grep '\.py$') | tar -x -C "$d"
  fi
  if [ "$strip" = 0 ]; then
    (cd "$d" && patch -p0 --dry-run -s -f < "$here/$patch" > /dev/null 2>&1)
  else
    (cd "$d" && git apply --check "$here/$patch" 2> /dev/null)
  fi
  if [ $? -eq 0 ]; then echo "ok       $patch @ $commit"; else echo "STALE    $patch does not apply to $commit"; fail=1; fi
  rm -rf "$d"
done < "$here/PATCHES.txt"
for p in $(cd "$here" && git ls-files '*.patch'); do
  case " $indexed " in *" $p "*) ;; *) echo "UNINDEXED $p"; fail=1 ;; esac
done
exit $fail
