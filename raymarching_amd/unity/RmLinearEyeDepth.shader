// RmLinearEyeDepth.shader -- writes the camera's depth texture as linear eye depth, one float per pixel: the value the
// reference's fragment program derives for every pixel before it marches,
//     Linear01Depth(tex2D(_CameraDepthTexture, uv).r) * _ProjectionParams.z        (Assets/Raymarching.shader:154)
// RaymarchingNative.cs blits this into an RFloat target, reads it back and passes it to rm_blit as scene_depth.
Shader "Hidden/RmLinearEyeDepth"
{
    SubShader
    {
        Cull Off ZWrite Off ZTest Always
        Pass
        {
            CGPROGRAM
            #pragma vertex vert_img
            #pragma fragment frag
            #include "UnityCG.cginc"

            sampler2D _CameraDepthTexture;

            float4 frag(v2f_img i) : SV_Target
            {
                float eye = Linear01Depth(SAMPLE_DEPTH_TEXTURE(_CameraDepthTexture, i.uv)) * _ProjectionParams.z;
                return float4(eye, 0, 0, 1);
            }
            ENDCG
        }
    }
}
