for rep in 1 2; do
for t in _r04 .; do n=$( [ $t = . ] && echo r05 || echo r04 )
  ( cd $t && timeout -k 10 300 python profiles/operator_path.py 136 2>/dev/null | grep -v amdgpu | sed "s/^/$n operator: /" )
done; done
